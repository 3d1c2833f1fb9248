#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02c
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
export SMMC_SEED=1 SMMC_VERBOSE=1
B=stock_market_monte_carlo_amd/bin
for PIN in whole 0 chunk whole 0 chunk; do
  echo "== SMMC_PIN_HOST=$PIN benchmark_mc_gpu 1 360 100000000"
  SMMC_PIN_HOST=$PIN timeout -k 10 120 $B/benchmark_mc_gpu 1 360 100000000 2>&1 | grep "smmc:\|All "
done > $OUT/cli_pin.txt 2>&1
cat $OUT/cli_pin.txt
unset SMMC_VERBOSE
bash tools/run_clis.sh > $OUT/cli_runs.txt 2>&1; tail -30 $OUT/cli_runs.txt
PROF_TAG=prof_r02 bash tools/profile_r02.sh 2>&1 | tail -5
