mkdir -p gpurun_out/r03f
for k in 1 2 3; do
  timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r03f/bench_try$k.json 2> /dev/null
  python tools/bench_line.py gpurun_out/r03f/bench_try$k.json
done
