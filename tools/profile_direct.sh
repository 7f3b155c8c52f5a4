#!/bin/bash
# Profiles the headline bench on the GPU box: one kernel-trace/stats run and four counter passes
# (counters never combined with sys/runtime traces).  Usage (from the repo root, on the box):
#   bash tools/profile_direct.sh gpurun_out/prof_r16
# The program under rocprofv3 is `python3 bench.py ...` directly (no env/bash hop; bench.py starts no child process once the GPU is up: clock and power come from sysfs, and --no-clock switches even that off here).
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/prof}")
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-clock --kernel-iters 2"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- \
    python3 "$REPO/bench.py" $ARGS > "$OUT/stats.json" 2> "$OUT/stats.err"
for group in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" \
             "sq:GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  name=${group%%:*}; counters=${group#*:}
  timeout -k 10 280 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$OUT/$name" -o run -- \
      python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-clock --kernel-iters 1 > "$OUT/$name.json" 2> "$OUT/$name.err"
  echo "pass $name done"
done
find "$OUT" -name "*.csv" | head -40
