#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/kd_comb
mkdir -p $OUT
cd $R
timeout -k 10 600 python -m pytest tests/test_keepdata_comb_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
ROUNDS=5 KD_SHAPES=4000000x360,1500000x1000,600000x360 timeout -k 10 600 python tools/kd_ab.py "SMMC_KEEPDATA_KERNEL=tile" "SMMC_KEEPDATA_KERNEL=comb" "SMMC_KEEPDATA_KERNEL=comb,SMMC_KEEPDATA_K=2" "SMMC_KEEPDATA_KERNEL=comb,SMMC_KEEPDATA_COMB_WAVES=12" "SMMC_KEEPDATA_KERNEL=comb,SMMC_KEEPDATA_COMB_ILP=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
