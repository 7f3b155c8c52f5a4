#!/bin/bash
# Counter passes over the Barnes-Hut walk at N = 2^20 (two-galaxy).  bash tools/profile_bh.sh <outdir> [walk form]
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/prof_bh}")
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for group in "sq1:GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
             "sq2:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU" \
             "sq3:SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_VALU_TRANS" \
             "sqc:SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ SQC_DCACHE_REQ_READ_16" \
             "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  name=${group%%:*}; counters=${group#*:}
  timeout -k 10 200 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$OUT/$name" -o run -- \
      python3 "$REPO/tools/bh_small_trace.py" ${NBH_PROFILE_N:-1048576} 4 two_galaxies $2 > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
  echo "pass $name done"
done
