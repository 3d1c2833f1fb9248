#!/bin/bash
# Development: A/B of a -D switch on the keepdata kernel: tools/kd_macro_ab.sh MACRO [kd_ab.py variants...]
M=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
for X in 0 1 0 1; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -x hip -shared -D$M=$X -I$R/include -I$R/stock_market_monte_carlo_amd/csrc -o $R/stock_market_monte_carlo_amd/libsmmc_hip.so $R/stock_market_monte_carlo_amd/csrc/smmc_kernels.hip $R/stock_market_monte_carlo_amd/csrc/smmc_stats_kernels.hip $R/stock_market_monte_carlo_amd/csrc/smmc_capi.cpp $R/stock_market_monte_carlo_amd/csrc/smmc_dropin.cpp || exit 1
  echo "== $M=$X"
  ROUNDS=${ROUNDS:-4} timeout -k 10 300 python3 $R/tools/kd_ab.py "$@" 2>&1 | grep -v amdgpu.ids || exit 1
done
