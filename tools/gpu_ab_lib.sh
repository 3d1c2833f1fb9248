#!/bin/bash
# Development: interleaved A/B of two builds of the library on one box:
#   tools/gpu_ab_lib.sh OUT_TAG LIB_A [LIB_B]   (LIB_B defaults to the product library)
# runs bench.py config 1 and tools/bench_keepdata.py against each, alternating, twice.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; mkdir -p $OUT
A=$2; B=${3:-$R/stock_market_monte_carlo_amd/libsmmc_hip.so}
cd $R
for X in A B A B; do
  L=$A; [ $X = B ] && L=$B
  echo "== $X $L"
  SMMC_LIB=$L timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])" || exit 1
  SMMC_LIB=$L timeout -k 10 200 python tools/bench_keepdata.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('keepdata', d['mode'], d['n_paths'], d['n_periods'], d['kernel_ms'], d['frac_of_8TBps'])" || exit 1
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
