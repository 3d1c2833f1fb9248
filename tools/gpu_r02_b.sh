#!/bin/bash
# Round-2 second GPU pass: the full -m gpu suite, the drop-in command lines with phase timers,
# keepdata baseline timings.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02b
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -15 $OUT/pytest_gpu.log
export SMMC_SEED=1 SMMC_VERBOSE=1
B=stock_market_monte_carlo_amd/bin
for PIN in 0 whole chunk 0 whole chunk; do
  echo "== SMMC_PIN_HOST=$PIN benchmark_mc_gpu 1 360 100000000"
  SMMC_PIN_HOST=$PIN timeout -k 10 120 $B/benchmark_mc_gpu 1 360 100000000 2>&1 | grep "smmc:\|All "
done > $OUT/cli_pin.txt 2>&1
cat $OUT/cli_pin.txt
timeout -k 10 300 python tools/bench_keepdata.py > $OUT/keepdata.jsonl 2> $OUT/keepdata.err; echo "keepdata rc=$?"; cat $OUT/keepdata.jsonl | cut -c1-200
