#!/bin/bash
# Counter passes over the spatial-hash force kernel at N = 4,194,304.  bash tools/profile_hash.sh <outdir> [force kernel]
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/prof_hash}")
REPO=$(pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for group in "sq1:GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" \
             "sq2:SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU" \
             "sq3:SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS"; do
  name=${group%%:*}; counters=${group#*:}
  timeout -k 10 200 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d "$OUT/$name" -o run -- \
      python3 "$REPO/tools/hash_trace.py" 4194304 4 $2 > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
  echo "pass $name done"
done
