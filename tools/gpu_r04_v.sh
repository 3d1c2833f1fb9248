#!/bin/bash
# Round 4, GPU pass V: copies of values_stats' LDS histogram (SMMC_STATS_HIST_COPIES: 8 ... 64), final values of configs[1].
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r04v; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
cat > /tmp/vs.py <<PY
import sys, json, torch, os
sys.path.insert(0, "$R")
import stock_market_monte_carlo_amd as S
e = S.Engine(0)
sim = S.Engine.make_sim(100_000_000, 360, S.MODE_GAUSSIAN, 7)
final = e.simulate(sim).final
big = torch.cat([final] * 10)
def timed(fn, reps):
    fn(); e.sync(); e.timing(True)
    for _ in range(reps): fn()
    ms, k = e.kernel_ms(); e.timing(False)
    return ms / max(k, 1)
out = {"copies": os.environ.get("SMMC_STATS_HIST_COPIES")}
out["final_1e8_us"] = round(timed(lambda: e.values_stats(final, 1000.0, 100, 0.0, 20000.0), 20) * 1e3, 2)
out["final_1e9_us"] = round(timed(lambda: e.values_stats(big, 1000.0, 100, 0.0, 20000.0), 10) * 1e3, 1)
out["final_1e8_nohist_us"] = round(timed(lambda: e.values_stats(final, 1000.0, 0, 0.0, 20000.0), 20) * 1e3, 2)
st = e.read_stats(e.values_stats(final, 1000.0, 100, 0.0, 20000.0))
out["hist_total"] = int(st.hist.sum()) + st.underflow + st.overflow
print(json.dumps(out))
PY
for round in 1 2; do
  for c in 16 8 32 48 64; do
    SMMC_STATS_HIST_COPIES=$c timeout -k 10 200 python /tmp/vs.py 2>/dev/null | tail -1
  done
done | tee $O/values_stats_copies.txt
