#!/bin/bash
# Round 4, GPU pass E: reference-stream trajectories with 8 / 4 / 2 / 1 rows per stream (tests, timing), the
# instruction-order and scalar-port probes (tools/ubench_ops.hip), the sustained-load clock of the kernels' own mix.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04e
mkdir -p $OUT
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > $OUT/build.log 2>&1; }
timeout -k 10 900 python -m pytest tests/test_ref_stream_gpu.py -m gpu -q -x > $OUT/pytest_ref.log 2>&1; echo "pytest ref rc=$?" | tee -a $OUT/pytest_ref.log
tail -4 $OUT/pytest_ref.log
timeout -k 10 400 python tools/bench_ref.py --traj > $OUT/bench_ref_traj.jsonl 2> $OUT/bench_ref_traj.err; echo "bench_ref traj rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04e/bench_ref_traj.jsonl"):
    d=json.loads(l); print("%-20s K=%-4s %9d x %4d  %.3f ms  %.2f TB/s  %.3f" % (d["case"], d["rows_per_stream"], d["n_paths"], d["n_periods"], d["kernel_ms"], d["TBps"], d["frac_of_8TBps"]))
PY
hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o $OUT/ubench_mix 2> $OUT/ubench_mix_build.log && timeout -k 10 300 $OUT/ubench_mix > $OUT/ubench_mix.txt 2>&1; echo "ubench_mix rc=$?"
grep sustained $OUT/ubench_mix.txt | cut -c1-220
hipcc -O3 --offload-arch=gfx950 tools/ubench_ops.hip -o $OUT/ubench_ops 2> $OUT/ubench_ops_build.log && timeout -k 10 300 $OUT/ubench_ops > $OUT/ubench_ops.jsonl 2>&1; echo "ubench_ops rc=$?"
python - <<'PY'
import json
rows=[]
for l in open("gpurun_out/r04e/ubench_ops.jsonl"):
    if l.startswith("{"):
        try: rows.append(json.loads(l))
        except ValueError: print("BAD LINE", l[:200])
for r in rows:
    if r["probe"].startswith(("order", "mix")): print("%-44s %-28s %dw  %.2f clk per 64th of the block" % (r["probe"], r["operands"], r["waves_per_simd"], r["clk_per_inst"]))
PY
rm -f $OUT/ubench_ops $OUT/ubench_mix
ls $OUT
