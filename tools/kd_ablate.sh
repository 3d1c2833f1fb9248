#!/bin/bash
# keepdata ablations (development): rebuilds the library with SMMC_KD_EXPERIMENT=0..3
for X in ${KD_EXPERIMENTS:-0 1 2 3}; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -x hip -shared -DSMMC_KD_EXPERIMENT=$X -Iinclude -Istock_market_monte_carlo_amd/csrc -o stock_market_monte_carlo_amd/libsmmc_hip.so stock_market_monte_carlo_amd/csrc/smmc_kernels.hip stock_market_monte_carlo_amd/csrc/smmc_stats_kernels.hip stock_market_monte_carlo_amd/csrc/smmc_capi.cpp stock_market_monte_carlo_amd/csrc/smmc_dropin.cpp || exit 1
  echo "experiment $X (0 full, 1 no global stores, 2 no draws, 3 compute only, 4 no store phase)"
  python3 tools/bench_keepdata.py 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-8s P=%-5d %.3f ms'%(d['mode'],d['n_periods'],d['kernel_ms']))" || exit 1
done
