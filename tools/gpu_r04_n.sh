#!/bin/bash
# Round 4, GPU pass N: what the bank conflicts of the Gaussian draw's two LDS gathers cost paths_kernel -- A/B of the product (B)
# against scratch builds whose gathers all read ONE entry (same instructions, a broadcast instead of a conflicted gather; wrong
# values, never the product): Z both gathers, ZR the radius cubic only, ZC the (cos, sin) pair only; P: conflict-free with DIFFERENT
# entries per lane (the low entry bits replaced by the lane number: +1 VALU per gather).  Interleaved on one box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04n
mkdir -p $OUT
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > $OUT/build.log 2>&1; }
D=$R/stock_market_monte_carlo_amd
one() {  # label lib bench-args...
  local L=$1 LIB=$2; shift 2
  SMMC_LIB=$LIB timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', '$*', '%.4g' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'clock %.3f' % d['valu']['held_clock_ghz'])"
}
for round in 1 2 3; do
  one B $D/libsmmc_hip.so --config 1; one Z $D/_build/libsmmc_hip_Z.so --config 1; one ZR $D/_build/libsmmc_hip_ZR.so --config 1; one ZC $D/_build/libsmmc_hip_ZC.so --config 1; one P $D/_build/libsmmc_hip_P.so --config 1
done 2>&1 | tee $OUT/ab_gather_conflicts.txt
for L in B Z; do
  LIB=$D/libsmmc_hip.so; [ $L = Z ] && LIB=$D/_build/libsmmc_hip_Z.so
  SMMC_LIB=$LIB timeout -k 10 200 python tools/bench_keepdata.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        if j['mode'] == 'gaussian': print('$L keepdata', j['n_paths'], j['n_periods'], j['kernel_ms'], j['GBps'])"
done 2>&1 | tee -a $OUT/ab_gather_conflicts.txt
