#!/bin/bash
# Round 4, GPU pass Q: the statistics record without per-call memsets (finalize_kernel folds the engine's bucket accumulator into
# the record and leaves it zero): the -m gpu suite, then the 1e6-path step and the statistics kernels against the build before
# (MEMSET: scratch build of the previous commit), interleaved on one box.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04q
mkdir -p $OUT
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > $OUT/build.log 2>&1; }
D=$R/stock_market_monte_carlo_amd
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
grep -q "pytest rc=0" $OUT/pytest_gpu.log || exit 1
for i in 1 2 3; do
for v in "product:$D/libsmmc_hip.so" "memset:$D/_build/libsmmc_hip_MEMSET.so"; do
  SMMC_LIB=${v#*:} timeout -k 10 300 python bench.py --config 0 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d.get('hbm_bound_kernels') or {}; print('config0 ${v%%:*}', '%.4g' % d['value'], 'us/step %.2f' % (d['ms_per_step']*1e3), 'kernel us %.2f' % (d['roofline']['kernel_ms']*1e3), 'values_stats us %.2f' % (h.get('values_stats', {}).get('kernel_ms', 0)*1e3))"
done; done 2>&1 | tee $OUT/config0_no_memset.txt
for v in "product:$D/libsmmc_hip.so" "memset:$D/_build/libsmmc_hip_MEMSET.so"; do
  SMMC_LIB=${v#*:} timeout -k 10 300 python bench.py --config 1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['hbm_bound_kernels']; print('config1 ${v%%:*}', '%.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'values_stats us %.2f' % (h['values_stats']['kernel_ms']*1e3), 'radix us %.2f' % (h['quartiles_radix_pass']['kernel_ms']*1e3))"
done 2>&1 | tee -a $OUT/config0_no_memset.txt
