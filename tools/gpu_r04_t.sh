#!/bin/bash
# Round 4, GPU pass T: values_stats, straight-line form (product) against the branchy form of the commit before (OLDSTATS),
# interleaved on one box: the bench line's hbm_bound_kernels at 1e8 final values, and 1e9 values.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r04t; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
D=$R/stock_market_monte_carlo_amd
cat > /tmp/vs.py <<PY
import sys, json, torch
sys.path.insert(0, "$R")
import stock_market_monte_carlo_amd as S
e = S.Engine(0)
sim = S.Engine.make_sim(100_000_000, 360, S.MODE_GAUSSIAN, 7)
final = e.simulate(sim).final
big = torch.cat([final] * 10)
uni = torch.rand(100_000_000, device="cuda:0") * 30000.0 - 5000.0   # a third of the values outside the buckets
def timed(fn, reps):
    fn(); e.sync(); e.timing(True)
    for _ in range(reps): fn()
    ms, k = e.kernel_ms(); e.timing(False)
    return ms / max(k, 1)
out = {}
out["final_1e8_us"] = round(timed(lambda: e.values_stats(final, 1000.0, 100, 0.0, 20000.0), 20) * 1e3, 2)
out["final_1e9_us"] = round(timed(lambda: e.values_stats(big, 1000.0, 100, 0.0, 20000.0), 10) * 1e3, 1)
out["final_1e8_nohist_us"] = round(timed(lambda: e.values_stats(final, 1000.0, 0, 0.0, 20000.0), 20) * 1e3, 2)
out["uniform_1e8_us"] = round(timed(lambda: e.values_stats(uni, 1000.0, 100, 0.0, 20000.0), 20) * 1e3, 2)
out["final_1e8_1000bins_us"] = round(timed(lambda: e.values_stats(final, 1000.0, 1000, 0.0, 20000.0), 20) * 1e3, 2)
print(json.dumps(out))
PY
for round in 1 2 3; do
  for v in "product:$D/libsmmc_hip.so" "oldstats:$D/_build/libsmmc_hip_OLDSTATS.so"; do
    echo -n "${v%%:*} "; SMMC_LIB=${v#*:} timeout -k 10 200 python /tmp/vs.py 2>/dev/null | tail -1
  done
done | tee $O/values_stats_forms.txt
