#!/bin/bash
# Everything the round's numbers come from, in one go on the GPU box (from the repo root):
#   bash tools/profile_round.sh gpurun_out/r02
# bench lines (default with extras, bh, hash, sharded hash on one rank), rocprofv3 kernel-trace/stats of the
# same commands, counter passes of the three force kernels, Barnes-Hut lane-participation histogram.
set -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/round}")
REPO=$(pwd)
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "bench default rc=$?"
python3 bench.py --workload bh --steps 50 --warmup 5 > "$OUT/bench_bh.json" 2> /dev/null; echo "bench bh rc=$?"
python3 bench.py --workload hash --steps 50 --warmup 5 > "$OUT/bench_hash.json" 2> /dev/null; echo "bench hash rc=$?"
python3 bench.py --workload hash --steps 50 --warmup 5 --force-sharded --no-cpu-baseline > "$OUT/bench_hash_sharded_1rank.json" 2> /dev/null; echo "bench hash sharded rc=$?"
bash tools/profile_direct.sh "$OUT/prof_direct" > "$OUT/prof_direct.log" 2>&1; echo "profile direct rc=$?"
python3 tools/pmc_summarise.py "$OUT/prof_direct" "direct_sym_kernel<16, false, true, true>" 1048576 > "$OUT/pmc_direct.json"
( cd /tmp && export TMPDIR=/tmp
  for w in bh hash; do
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$w" -o run -- \
        python3 "$REPO/bench.py" --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-clock > /dev/null 2>&1
    echo "stats $w rc=$?"
  done )
bash tools/profile_bh.sh "$OUT/prof_bh" > "$OUT/prof_bh.log" 2>&1; echo "profile bh rc=$?"
python3 tools/pmc_kernel_summarise.py "$OUT/prof_bh" "bh_traverse_pair_kernel<false, 1>" "two_galaxies N=1048576 theta=0.5 eps=0.05" > "$OUT/pmc_bh.json"
bash tools/profile_hash.sh "$OUT/prof_hash" > "$OUT/prof_hash.log" 2>&1; echo "profile hash rc=$?"
python3 tools/pmc_kernel_summarise.py "$OUT/prof_hash" hash_cell_force "uniform box N=4194304, cell = cutoff = 1" > "$OUT/pmc_hash.json"
python3 tools/bh_mask_hist.py two_galaxies 2>/dev/null > "$OUT/bh_lane_participation.txt"
python3 tools/bh_mask_hist.py plummer 2>/dev/null >> "$OUT/bh_lane_participation.txt"
python3 tools/bh_depth_sweep.py 2>/dev/null > "$OUT/bh_depth_sweep.txt"
python3 tools/bh_form_sweep.py 262144 524288 1048576 4194304 2>/dev/null > "$OUT/bh_walk_forms.txt"
python3 tools/bh_build_time.py 1048576 4194304 2>/dev/null > "$OUT/bh_build_time.txt"
python3 tools/bh_split_vs_pair.py 2>/dev/null > "$OUT/bh_split_vs_pair.txt"
python3 -m pytest tests/test_direct_gpu.py tests/test_barnes_hut_gpu.py -q -s -k "headline or config4" > "$OUT/headline_parity_and_config4_drift.log" 2>&1
echo done
