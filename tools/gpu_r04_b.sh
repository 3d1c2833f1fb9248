#!/bin/bash
# Round 4, GPU pass B: the new parity tests on the rebuilt library, the per-opcode issue-rate table by operand kind
# (tools/ubench_ops.hip), the 1e6-path step.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04b
mkdir -p $OUT
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > $OUT/build.log 2>&1; }
timeout -k 10 1000 python -m pytest tests/test_gaussian_reference_gpu.py tests/test_full_size_configs_gpu.py tests/test_finalize_gpu.py tests/test_bench_gpu.py tests/test_group_gpu.py tests/test_host_pipeline_gpu.py -m gpu -q -s -x > $OUT/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $OUT/pytest_new.log
tail -25 $OUT/pytest_new.log
hipcc -O3 --offload-arch=gfx950 tools/ubench_ops.hip -o $OUT/ubench_ops 2> $OUT/ubench_ops_build.log && timeout -k 10 300 $OUT/ubench_ops > $OUT/ubench_ops.jsonl 2>&1; echo "ubench_ops rc=$?"
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r04b/ubench_ops.jsonl") if l.startswith("{")]
for r in rows:
    if r["waves_per_simd"]==8: print("%-18s %-24s %.3f Ginst/s/SIMD  %.2f clk" % (r["probe"], r["operands"], r["ginst_per_s_per_simd"], r["clk_per_inst"]))
PY
for i in 1 2 3; do timeout -k 10 300 python bench.py --config 0 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config0', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done
SMMC_SMALL_LAUNCH_ROUNDS=0 timeout -k 10 300 python bench.py --config 0 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config0 uncapped', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
SMMC_FINALIZE=launch SMMC_SMALL_LAUNCH_ROUNDS=0 timeout -k 10 300 python bench.py --config 0 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config0 r03 form', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
rm -f $OUT/ubench_ops
ls $OUT
