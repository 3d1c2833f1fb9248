#!/bin/bash
# Round 4, GPU pass F: which of the two operand-kind changes costs the table kernel its 0.9 % (interleaved A/B of four
# builds on one box: A = before both, B = the product, T1 = round keys scalar in table mode, T2 = table gathers through the
# extern __shared__ symbol), then the whole -m gpu suite.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04f
mkdir -p $OUT
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > $OUT/build.log 2>&1; }
D=$R/stock_market_monte_carlo_amd
one() {  # label lib bench-args...
  local L=$1 LIB=$2; shift 2
  SMMC_LIB=$LIB timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', '$*', '%.4g' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])"
}
for round in 1 2 3; do
  one A $D/_build/libsmmc_hip_A.so --config 2; one B $D/libsmmc_hip.so --config 2; one T1 $D/_build/libsmmc_hip_T1.so --config 2; one T2 $D/_build/libsmmc_hip_T2.so --config 2
done 2>&1 | tee $OUT/ab_table.txt
for i in 1 2; do
for v in "default:" "uncapped:SMMC_SMALL_LAUNCH_ROUNDS=0" "r03form:SMMC_FINALIZE=launch SMMC_SMALL_LAUNCH_ROUNDS=0" "launch+cap:SMMC_FINALIZE=launch"; do
  env ${v#*:} timeout -k 10 300 python bench.py --config 0 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config0 ${v%%:*}', '%.4g' % d['value'], 'us/step %.2f' % (d['ms_per_step']*1e3), 'kernel us %.2f' % (d['roofline']['kernel_ms']*1e3), 'clock', d['valu']['held_clock_ghz'])"
done; done 2>&1 | tee $OUT/config0_variants.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config1', d['value'], d['roofline']['kernel_ms'], d['valu'])"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?" | tee -a $OUT/pytest_gpu.log
tail -6 $OUT/pytest_gpu.log
ls $OUT
