#!/bin/bash
# Round 4, GPU pass R: workgroups per CU of the statistics / radix kernels at 1e8 and 1e9 values (SMMC_STATS_BLOCKS_PER_CU).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04r; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
for round in 1 2; do
for b in 2 4 8 16; do
  SMMC_STATS_BLOCKS_PER_CU=$b timeout -k 10 200 python - <<PY
import os, sys, json, torch
sys.path.insert(0, "$R")
import stock_market_monte_carlo_amd as S
e = S.Engine(0)
out = {"blocks_per_cu": $b}
for n in (100_000_000, 1_000_000_000):
    v = torch.rand(n, device="cuda:0") * 9000.0 + 500.0
    def timed(fn, reps):
        fn(); e.sync(); e.timing(True)
        for _ in range(reps): fn()
        ms, k = e.kernel_ms(); e.timing(False)
        return ms / max(k, 1)
    ms = timed(lambda: e.values_stats(v, 1000.0, 100, 0.0, 20000.0), 10)
    out[f"values_stats_{n:.0e}"] = round(4.0 * n / ms / 1e6)
    ms = timed(lambda: e.quartiles(v), 5)
    out[f"radix_pass_{n:.0e}"] = round(4.0 * n / ms / 1e6)
    del v
print(json.dumps(out))
PY
done; done 2>&1 | tee $O/stats_blocks_per_cu.txt
