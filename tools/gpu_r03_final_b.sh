#!/bin/bash
# Round-3 final GPU pass, part B: drop-in command lines with phase timers, cold start, group and
# host-pipeline micro-benchmarks, the generic reference-stream kernel's sweep, then the rocprofv3 passes.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03f
mkdir -p $OUT
cd $R
{
echo "# drop-in command lines on the GPU box (one MI355X, SMMC_SEED=1); SMMC_VERBOSE phase timers on stderr."
export SMMC_SEED=1 SMMC_VERBOSE=1
B=stock_market_monte_carlo_amd/bin
for PIN in whole 0 whole 0; do
  echo "== SMMC_PIN_HOST=$PIN benchmark_mc_gpu 1 360 100000000"
  SMMC_PIN_HOST=$PIN timeout -k 10 120 $B/benchmark_mc_gpu 1 360 100000000 2>&1 | grep "smmc:\|All \|mean"
done
echo "== SMMC_DEVICE_MAP=0,0,0 benchmark_mc_gpu 3 360 100000000 (three shards, one GPU)"
SMMC_DEVICE_MAP=0,0,0 timeout -k 10 120 $B/benchmark_mc_gpu 3 360 100000000 2>&1 | grep "smmc:\|All \|mean"
echo "== SMMC_STREAM=ref SMMC_SEED=1000 benchmark_mc_gpu 1 360 100000000 (the reference CPU engine's own stream)"
SMMC_STREAM=ref SMMC_SEED=1000 timeout -k 10 120 $B/benchmark_mc_gpu 1 360 100000000 2>&1 | grep "smmc:\|All \|mean\|count"
echo "== SMMC_STREAM=ref SMMC_SEED=1000 benchmark_mc_cpu_v2 360 1000000 (BASELINE configs[0] size)"
SMMC_STREAM=ref SMMC_SEED=1000 timeout -k 10 120 $B/benchmark_mc_cpu_v2 360 1000000 2>&1 | grep "smmc:\|All \|mean\|count"
echo "== the reference's own examples/benchmark_mc_gpu.cpp, compiled unmodified (oracle/_ref), LOCPATH=oracle/_ref/locale"
LOCPATH=$R/oracle/_ref/locale timeout -k 10 120 oracle/_ref/benchmark_mc_gpu 1 360 100000000 2>&1 | grep "smmc:\|All \|mean\|count"
echo
unset SMMC_VERBOSE
bash tools/run_clis.sh 2>&1
} > $OUT/cli_runs.txt 2>&1
grep "All \|engines up" $OUT/cli_runs.txt | head -30
timeout -k 10 300 bash tools/cold_start.sh > $OUT/cold_start.txt 2>&1; echo "cold start rc=$?"
g++ -O2 -std=c++17 -Iinclude tools/ubench_group.cpp -o /tmp/ubench_group -Lstock_market_monte_carlo_amd -lsmmc_hip -Wl,-rpath,$PWD/stock_market_monte_carlo_amd || exit 1
(timeout -k 10 120 /tmp/ubench_group 1 1000000 36 20; timeout -k 10 120 /tmp/ubench_group 1 100000000 360 10) > $OUT/ubench_group.txt 2>&1; echo "ubench_group rc=$?"
timeout -k 10 300 python tools/bench_host_chunks.py > $OUT/host_chunks.jsonl 2>/dev/null; echo "host chunks rc=$?"
timeout -k 10 400 bash tools/ref_generic_sweep.sh > $OUT/ref_generic_sweep.txt 2>&1; echo "generic sweep rc=$?"
timeout -k 10 200 python tools/bench_stats.py 100000000 > $OUT/bench_stats.jsonl 2>/dev/null; timeout -k 10 200 python tools/bench_stats.py 1000000000 >> $OUT/bench_stats.jsonl 2>/dev/null; echo "stats rc=$?"
PROF_TAG=r03f/prof timeout -k 10 500 bash tools/profile_r03.sh > $OUT/profile.log 2>&1; echo "profile rc=$?"; tail -3 $OUT/profile.log
