#!/bin/bash
# Development: counter passes over the keepdata kernel (tools/kd_one.py)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/kd_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
P=0
while read -r SET; do
  P=$((P+1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pass$P -- python3 $R/tools/kd_one.py > $OUT/pass$P.log 2>&1 || { tail -5 $OUT/pass$P.log; exit 1; }
done <<SETS
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE
WRITE_SIZE
SETS
cd $R
python3 $R/tools/pmc_summary.py $OUT/pass* | grep -A9 "keepdata" 
