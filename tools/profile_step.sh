#!/bin/bash
# Counter passes over the SAME command the bench records use: bench.py --workload <bh|hash|direct> (BASELINE configs 4 /
# 5 / 2-3 with the bench's own parameters), one rocprofv3 --pmc pass per counter group (counters only with
# --kernel-trace, never with other trace domains).   bash tools/profile_step.sh <workload> <outdir> [extra bench args]
set -e -o pipefail
WL=${1:-bh}
OUT=$(realpath -m "${2:-gpurun_out/prof_$WL}")
shift 2 || true
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for group in "sq1:GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
             "sq2:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU" \
             "sq3:SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VALU_TRANS SQ_LDS_BANK_CONFLICT" \
             "sqc:SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ" \
             "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=${group%%:*}; counters=${group#*:}
  timeout -k 10 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$OUT/$name" -o run -- \
      python3 "$REPO/bench.py" --workload "$WL" --steps 4 --warmup 2 --no-cpu-baseline --no-extra --no-clock --kernel-iters 1 "$@" > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
  echo "pass $name done"
done
