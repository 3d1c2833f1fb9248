#!/bin/bash
# Development: builds an experimental variant of the library WITHOUT touching the product
# stock_market_monte_carlo_amd/libsmmc_hip.so:   tools/variant_build.sh TAG [-DMACRO=1 ...]
# -> stock_market_monte_carlo_amd/_build/libsmmc_hip_TAG.so ; run anything against it with SMMC_LIB=<that path>.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
mkdir -p $R/stock_market_monte_carlo_amd/_build
OUT=$R/stock_market_monte_carlo_amd/_build/libsmmc_hip_$TAG.so
C=$R/stock_market_monte_carlo_amd/csrc
hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -x hip -shared "$@" \
  -I$R/include -I$C -o $OUT $C/smmc_kernels.hip $C/smmc_stats_kernels.hip $C/smmc_capi.cpp $C/smmc_dropin.cpp || exit 1
echo $OUT
