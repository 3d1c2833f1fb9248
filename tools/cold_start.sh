#!/bin/bash
# Cold start of the drop-in command line, phase by phase (GPU box): `benchmark_mc_gpu 1 360 100000000` in fresh
# processes with SMMC_VERBOSE=2 (engine creation and the first simulate_to_host print cumulative phase
# times), once per pinning policy, and the C-ABI calls timed one by one (tools/ubench_startup.cpp).
export SMMC_SEED=1
B=stock_market_monte_carlo_amd/bin
g++ -O2 -std=c++17 -Iinclude tools/ubench_startup.cpp -o /tmp/ubench_startup -Lstock_market_monte_carlo_amd -lsmmc_hip -Wl,-rpath,$PWD/stock_market_monte_carlo_amd || exit 1
for rep in 1 2 3; do
  for pol in whole chunk 0; do
    echo "== run $rep: SMMC_PIN_HOST=$pol SMMC_VERBOSE=2 benchmark_mc_gpu 1 360 100000000"
    SMMC_PIN_HOST=$pol SMMC_VERBOSE=2 $B/benchmark_mc_gpu 1 360 100000000 2>&1 | grep -v "^count_below\|^n_periods\|^mean" || exit 1
  done
done
for rep in 1 2 3; do echo "== ubench_startup run $rep"; /tmp/ubench_startup || exit 1; done
