#!/bin/bash
# Round 4, GPU pass W: the quartiles CALL (ranks up, three passes and picks, five values back) without the per-pass memsets and
# without the synchronisation after the upload (product) against the commit before (OLDQ), interleaved; statistics tests first.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r04w; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
D=$R/stock_market_monte_carlo_amd
timeout -k 10 900 python -m pytest tests/test_stats_gpu.py tests/test_dropin_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x > $O/pytest_stats.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_stats.log
tail -3 $O/pytest_stats.log
grep -q "pytest rc=0" $O/pytest_stats.log || exit 1
cat > /tmp/qc.py <<PY
import sys, json, time, torch
sys.path.insert(0, "$R")
import stock_market_monte_carlo_amd as S
e = S.Engine(0)
sim = S.Engine.make_sim(100_000_000, 360, S.MODE_GAUSSIAN, 7)
final = e.simulate(sim).final
out = {}
for n in (10_000, 1_000_000, 10_000_000, 100_000_000):
    v = final[:n]
    q0 = e.quartiles(v)
    reps = 200 if n <= 1_000_000 else 30
    t0 = time.perf_counter()
    for _ in range(reps): q = e.quartiles(v)
    out[f"call_us_{n:.0e}"] = round((time.perf_counter() - t0) / reps * 1e6, 1)
    assert (q == q0).all()
print(json.dumps(out))
PY
for round in 1 2 3; do
  for v in "product:$D/libsmmc_hip.so" "before:$D/_build/libsmmc_hip_OLDQ.so"; do
    echo -n "${v%%:*} "; SMMC_LIB=${v#*:} timeout -k 10 200 python /tmp/qc.py 2>/dev/null | tail -1
  done
done | tee $O/quartiles_call.txt
