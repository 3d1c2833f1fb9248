#!/bin/bash
# Round 4, GPU pass S: radix passes 1 / 2 with the group table (product) against the select chain (SMMC_RADIX_MATCH=chain),
# per pass from the rocprofv3 kernel trace of quartiles() at 1e8 and 1e9 values; before that the statistics tests.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r04s; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
timeout -k 10 900 python -m pytest tests/test_stats_gpu.py tests/test_dropin_gpu.py tests/test_fuzz_gpu.py tests/test_ref_stream_gpu.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_stats.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_stats.log
tail -3 $O/pytest_stats.log
grep -q "pytest rc=0" $O/pytest_stats.log || exit 1
cat > /tmp/quart.py <<PY
import sys, json, torch
sys.path.insert(0, "$R")
import stock_market_monte_carlo_amd as S
e = S.Engine(0)
sim = S.Engine.make_sim(100_000_000, 360, S.MODE_GAUSSIAN, 7)
final = e.simulate(sim).final
for n in (100_000_000,):
    v = final[:n]
    for _ in range(12): q = e.quartiles(v)
    e.sync()
big = torch.cat([final] * 10)
for _ in range(6): q = e.quartiles(big)
e.sync()
print(q)
PY
export TMPDIR=/tmp
cd /tmp
for form in table chain; do
  if [ $form = chain ]; then export SMMC_RADIX_MATCH=chain; else unset SMMC_RADIX_MATCH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$form -- python3 /tmp/quart.py > $O/trace_$form.log 2>&1 || { tail -5 $O/trace_$form.log; exit 1; }
  F=$(find $O/trace_$form -name "*kernel_trace.csv" | head -1)
  python3 - "$F" $form <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    if "radix_hist_kernel" in name:
        by[name.split("radix_hist_kernel")[1][:3]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(by):
    d = by[k]; small = sorted(d[:12])[2:-2]; large = sorted(d[12:])[1:-1]
    print(sys.argv[2], "pass", k, "1e8 values: %.1f us" % (sum(small) / len(small) / 1e3), "  1e9 values: %.1f us" % (sum(large) / len(large) / 1e3))
PY
done 2>&1 | tee $O/radix_match_forms.txt
rm -rf $O/trace_table $O/trace_chain
unset SMMC_RADIX_MATCH
cd $R
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['hbm_bound_kernels']['quartiles_radix_pass'])); print(json.dumps(d['hbm_bound_kernels']['values_stats']))" | tee -a $O/radix_match_forms.txt
