#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export SMMC_KEEPDATA_KERNEL=comb SMMC_KEEPDATA_K=1
for X in 0 1 2 3 4; do
  for M in gaussian table; do
    echo -n "X=$X $M: "; SMMC_LIB=$R/stock_market_monte_carlo_amd/_build/libsmmc_hip_x$X.so KD_MODE=$M REPS=12 timeout -k 10 120 python tools/kd_one.py 2>/dev/null | tail -1
  done
done
