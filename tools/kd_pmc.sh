#!/bin/bash
# Development: counter passes over the keepdata kernels (tools/kd_one.py); environment selects the
# kernel and mode (SMMC_KEEPDATA_KERNEL, SMMC_KEEPDATA_K, KD_MODE, KD_P, KD_N); TAG names the output.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/kd_pmc_${TAG:-run}
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
WRITE_SIZE GRBM_GUI_ACTIVE
SETS
cd $R
python3 $R/tools/pmc_summary.py $OUT/pass* | grep -A10 "keepdata" > $OUT/summary.txt
cat $OUT/summary.txt
