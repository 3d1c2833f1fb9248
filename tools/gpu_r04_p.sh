#!/bin/bash
# Round-4 GPU pass P (the final tree): smoke, the whole -m gpu suite, bench.py presets (config 1 with the CPU
# baseline), the reference stream, the 2-rank launcher rehearsals, the HBM-bound neighbours.
# Part B (tools/gpu_r03_final_b.sh): command lines, cold start, group / host-pipeline micro-benchmarks, rocprofv3.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04p
mkdir -p $OUT
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > $OUT/build.log 2>&1; }
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
grep -q "pytest rc=0" $OUT/pytest_gpu.log || exit 1
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > $OUT/bench_config1.json 2> $OUT/bench_config1.err; echo "config1 rc=$?"
for c in 0 2 3; do timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err; echo "config$c rc=$?"; done
timeout -k 10 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_config4.json 2> $OUT/bench_config4.err; echo "config4 rc=$?"
timeout -k 10 300 python bench.py --config 4 --total-paths 125000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config4_share.json 2> $OUT/bench_config4_share.err; echo "config4 share rc=$?"
timeout -k 10 300 python bench.py --config 3 --total-paths 125000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config3_share.json 2> $OUT/bench_config3_share.err; echo "config3 share rc=$?"
timeout -k 10 300 python bench.py --stream ref --outputs final --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_stream_ref.json 2> $OUT/bench_stream_ref.err; echo "stream ref rc=$?"
timeout -k 10 300 python bench.py --stream ref --outputs all --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_stream_ref_all.json 2> $OUT/bench_stream_ref_all.err; echo "stream ref all rc=$?"
timeout -k 10 300 python bench.py --stream ref --config 0 --mode table --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_stream_ref_config0.json 2> $OUT/bench_stream_ref_config0.err; echo "stream ref config0 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 1 > $OUT/bench_gpus2_gloo.json 2> $OUT/bench_gpus2_gloo.err; echo "gpus2 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --config 3 --steps 3 --warmup 1 > $OUT/bench_gpus2_gloo_config3.json 2> $OUT/bench_gpus2_gloo_config3.err; echo "gpus2 c3 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --config 4 --total-paths 250000000 --steps 2 --warmup 1 > $OUT/bench_gpus2_gloo_config4.json 2> $OUT/bench_gpus2_gloo_config4.err; echo "gpus2 c4 rc=$?"
timeout -k 10 300 python3 bench.py --rehearse-rccl --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_rehearse_rccl.json 2> $OUT/bench_rehearse_rccl.err; echo "rehearse rccl rc=$?"
timeout -k 10 300 python tools/bench_keepdata.py > $OUT/bench_keepdata.jsonl 2> /dev/null; echo "keepdata rc=$?"
timeout -k 10 300 python tools/bench_ref.py > $OUT/bench_ref.jsonl 2> /dev/null; echo "bench_ref rc=$?"
timeout -k 10 300 python tools/bench_ref.py --traj > $OUT/bench_ref_traj.jsonl 2> /dev/null; echo "bench_ref traj rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04p/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], "%.4g"%d["value"], "ms/step %.2f"%d["ms_per_step"], "ranks", d["ranks"], d["scaling"], "kernel_ms %.3f"%d["roofline"]["kernel_ms"], "valu %.3f"%d["valu"]["frac"], "traffic", d["roofline"]["traffic"])
PY
cut -c1-150 $OUT/bench_keepdata.jsonl $OUT/bench_ref.jsonl
