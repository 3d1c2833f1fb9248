#!/bin/bash
# Round 4, GPU pass D: interleaved A/B on ONE box of the library before (A: _build/libsmmc_hip_A.so, commit 884a8d2) and
# after (B: the product) the operand-kind changes; the reference-stream trajectory tests on the comb writer and its
# timing; the per-opcode table with the explicit-register bank probes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04d
mkdir -p $OUT
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > $OUT/build.log 2>&1; }
A=$R/stock_market_monte_carlo_amd/_build/libsmmc_hip_A.so
B=$R/stock_market_monte_carlo_amd/libsmmc_hip.so
one() {  # label lib bench-args...
  local L=$1 LIB=$2; shift 2
  SMMC_LIB=$LIB timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', '$*', '%.4g' % d['value'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])"
}
for round in 1 2 3; do
  one A $A --config 1; one B $B --config 1
  one A $A --config 2; one B $B --config 2
  one A $A --stream ref --outputs final; one B $B --stream ref --outputs final
done 2>&1 | tee $OUT/ab_operands.txt
timeout -k 10 900 python -m pytest tests/test_ref_stream_gpu.py tests/test_dropin_gpu.py -m gpu -q -x > $OUT/pytest_ref.log 2>&1; echo "pytest ref rc=$?" | tee -a $OUT/pytest_ref.log
tail -6 $OUT/pytest_ref.log
timeout -k 10 300 python tools/bench_ref.py --traj > $OUT/bench_ref_traj.jsonl 2> $OUT/bench_ref_traj.err; echo "bench_ref traj rc=$?"
cut -c1-200 $OUT/bench_ref_traj.jsonl
hipcc -O3 --offload-arch=gfx950 tools/ubench_ops.hip -o $OUT/ubench_ops 2> $OUT/ubench_ops_build.log && timeout -k 10 300 $OUT/ubench_ops > $OUT/ubench_ops.jsonl 2>&1; echo "ubench_ops rc=$?"
python - <<'PY'
import json
rows=[]
for l in open("gpurun_out/r04d/ubench_ops.jsonl"):
    if l.startswith("{"):
        try: rows.append(json.loads(l))
        except ValueError: print("BAD LINE", l[:200])
for r in rows[96:]:
    if r["waves_per_simd"]==8: print("%-18s %-48s %.3f Ginst/s/SIMD  %.2f clk" % (r["probe"], r["operands"], r["ginst_per_s_per_simd"], r["clk_per_inst"]))
PY
rm -f $OUT/ubench_ops
ls $OUT
