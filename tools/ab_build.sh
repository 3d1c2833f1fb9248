#!/bin/bash
# A/B of a -D switch on the paths kernels (development): tools/ab_build.sh MACRO
M=$1
for X in 0 1 0 1; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -x hip -shared -D$M=$X -Iinclude -Istock_market_monte_carlo_amd/csrc -o stock_market_monte_carlo_amd/libsmmc_hip.so stock_market_monte_carlo_amd/csrc/smmc_kernels.hip stock_market_monte_carlo_amd/csrc/smmc_stats_kernels.hip stock_market_monte_carlo_amd/csrc/smmc_capi.cpp stock_market_monte_carlo_amd/csrc/smmc_dropin.cpp || exit 1
  for m in gaussian table; do
    echo -n "$M=$X $m: "; python3 bench.py --mode $m --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4g paths/s kernel_ms=%.3f'%(d['value'], d['roofline']['kernel_ms']))" || exit 1
  done
done
