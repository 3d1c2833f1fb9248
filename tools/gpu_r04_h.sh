#!/bin/bash
# Round 4, GPU pass H: the two-level fold inside the launch -- its tests, the 1e6-path step in both forms.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04h
mkdir -p $OUT
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > $OUT/build.log 2>&1; }
timeout -k 10 600 python -m pytest tests/test_finalize_gpu.py tests/test_gpu_parity.py tests/test_stats_gpu.py -m gpu -q -x > $OUT/pytest_finalize.log 2>&1; echo "pytest finalize rc=$?" | tee -a $OUT/pytest_finalize.log
tail -4 $OUT/pytest_finalize.log
grep -q "pytest finalize rc=0" $OUT/pytest_finalize.log || exit 1
for i in 1 2 3; do
for v in "fused:" "r03form:SMMC_FINALIZE=launch"; do
  env ${v#*:} timeout -k 10 300 python bench.py --config 0 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config0 ${v%%:*}', '%.4g' % d['value'], 'us/step %.2f' % (d['ms_per_step']*1e3), 'kernel us %.2f' % (d['roofline']['kernel_ms']*1e3), 'clock %.3f' % d['valu']['held_clock_ghz'])"
done; done 2>&1 | tee $OUT/config0_variants.txt
for c in 1 2; do timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config$c', '%.4g' % d['value'], d['roofline']['kernel_ms'], d['valu']['held_clock_ghz'])"; done
ls $OUT
