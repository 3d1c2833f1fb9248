#!/bin/bash
# Round 4, GPU pass C: after the operand-kind changes (round keys and logic-op constants in VGPRs, plain shifts for table
# addresses): the whole -m gpu suite, the bench presets, the reference stream, the extended per-opcode table.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04c
mkdir -p $OUT
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > $OUT/build.log 2>&1; }
for c in 1 2; do timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config$c', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['valu']['frac'])"; done
timeout -k 10 300 python bench.py --stream ref --outputs final --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ref360', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout -k 10 300 python bench.py --stream ref --outputs final --periods 1000 --paths-per-gpu 20000000 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ref1000', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout -k 10 300 python bench.py --config 0 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config0', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?" | tee -a $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
hipcc -O3 --offload-arch=gfx950 tools/ubench_ops.hip -o $OUT/ubench_ops 2> $OUT/ubench_ops_build.log && timeout -k 10 300 $OUT/ubench_ops > $OUT/ubench_ops.jsonl 2>&1; echo "ubench_ops rc=$?"
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r04c/ubench_ops.jsonl") if l.startswith("{")]
for r in rows:
    if r["waves_per_simd"]==8 and rows.index(r) >= 96: print("%-18s %-44s %.3f Ginst/s/SIMD  %.2f clk" % (r["probe"], r["operands"], r["ginst_per_s_per_simd"], r["clk_per_inst"]))
PY
rm -f $OUT/ubench_ops
ls $OUT
