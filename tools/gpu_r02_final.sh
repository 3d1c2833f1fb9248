#!/bin/bash
# Round-2 final GPU pass: smoke, the whole -m gpu suite, drop-in command lines with phase timers,
# bench.py presets, the 2-rank launcher rehearsal, rocprofv3 profile of the judged command.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02f
mkdir -p $OUT
cd $R
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -4 $OUT/pytest_gpu.log
{
echo "# drop-in command lines on the GPU box (one MI355X, SMMC_SEED=1); SMMC_VERBOSE phase timers on stderr."
echo "# Round 1 (profiles/r01/cli_runs.txt): benchmark_mc_gpu 1 360 100000000 0.233 s warm, of which ~0.12 s sizing the result vector,"
echo "# 0.035 s engine start-up, 0.047-0.08 s simulate + pageable copy."
export SMMC_SEED=1 SMMC_VERBOSE=1
B=stock_market_monte_carlo_amd/bin
for PIN in whole 0 whole 0; do
  echo "== SMMC_PIN_HOST=$PIN benchmark_mc_gpu 1 360 100000000"
  SMMC_PIN_HOST=$PIN timeout -k 10 120 $B/benchmark_mc_gpu 1 360 100000000 2>&1 | grep "smmc:\|All \|mean"
done
echo "== SMMC_DEVICE_MAP=0,0,0 benchmark_mc_gpu 3 360 100000000 (three shards, one GPU)"
SMMC_DEVICE_MAP=0,0,0 timeout -k 10 120 $B/benchmark_mc_gpu 3 360 100000000 2>&1 | grep "smmc:\|All \|mean"
echo "== the reference's own examples/benchmark_mc_gpu.cpp, compiled unmodified (oracle/_ref), LOCPATH=oracle/_ref/locale"
LOCPATH=$R/oracle/_ref/locale timeout -k 10 120 oracle/_ref/benchmark_mc_gpu 1 360 100000000 2>&1 | grep "smmc:\|All \|mean\|count"
echo
unset SMMC_VERBOSE
bash tools/run_clis.sh 2>&1
} > $OUT/cli_runs.txt 2>&1
grep "All \|engines up\|shard 0" $OUT/cli_runs.txt | head -30
timeout -k 10 400 python bench.py --steps 10 --warmup 2 > $OUT/bench_config1.json 2> $OUT/bench_config1.err; echo "config1 rc=$?"
for c in 0 2 3; do timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err; echo "config$c rc=$?"; done
timeout -k 10 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_config4.json 2> $OUT/bench_config4.err; echo "config4 rc=$?"
timeout -k 10 300 python bench.py --config 4 --total-paths 125000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config4_share.json 2> $OUT/bench_config4_share.err; echo "config4 share rc=$?"
timeout -k 10 300 python bench.py --config 3 --total-paths 125000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config3_share.json 2> $OUT/bench_config3_share.err; echo "config3 share rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 1 > $OUT/bench_gpus2_gloo.json 2> $OUT/bench_gpus2_gloo.err; echo "gpus2 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --config 3 --steps 3 --warmup 1 > $OUT/bench_gpus2_gloo_config3.json 2> $OUT/bench_gpus2_gloo_config3.err; echo "gpus2 c3 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --config 4 --total-paths 250000000 --steps 2 --warmup 1 > $OUT/bench_gpus2_gloo_config4.json 2> $OUT/bench_gpus2_gloo_config4.err; echo "gpus2 c4 rc=$?"
timeout -k 10 300 python3 bench.py --rehearse-rccl --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_rehearse_rccl.json 2> $OUT/bench_rehearse_rccl.err; echo "rehearse rccl rc=$?"
timeout -k 10 300 python tools/bench_keepdata.py > $OUT/bench_keepdata.jsonl 2> /dev/null; echo "keepdata rc=$?"
PROF_TAG=prof_r02f bash tools/profile_r02.sh 2>&1 | tail -3
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r02f/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f.split("/")[-1], "%.4g"%d["value"], "ms/step %.2f"%d["ms_per_step"], "ranks", d["ranks"], d["scaling"], "kernel_ms %.3f"%d["roofline"]["kernel_ms"], "valu %.3f"%d["valu"]["frac"], "traffic", d["roofline"]["traffic"])
PY
cut -c1-130 $OUT/bench_keepdata.jsonl
