#!/bin/bash
# Round-2 first GPU pass: parity tests, host-registration microbenchmark, bench.py presets, the
# 2-rank rehearsal of bench.py's own launcher, drop-in command lines with/without pinning.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02a
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
hipcc -O2 -o /tmp/ubench_hostreg tools/ubench_hostreg.cpp && timeout -k 10 120 /tmp/ubench_hostreg > $OUT/ubench_hostreg.txt 2>&1; cat $OUT/ubench_hostreg.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > $OUT/bench_config1.json 2> $OUT/bench_config1.err; echo "config1 rc=$?"
timeout -k 10 300 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_config2.json 2> $OUT/bench_config2.err; echo "config2 rc=$?"
timeout -k 10 300 python bench.py --config 3 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config3.json 2> $OUT/bench_config3.err; echo "config3 rc=$?"
timeout -k 10 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_config4.json 2> $OUT/bench_config4.err; echo "config4 rc=$?"
timeout -k 10 300 python bench.py --config 4 --total-paths 125000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config4_share.json 2> $OUT/bench_config4_share.err; echo "config4 share rc=$?"
timeout -k 10 300 python bench.py --config 3 --total-paths 125000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config3_share.json 2> $OUT/bench_config3_share.err; echo "config3 share rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 5 --warmup 1 > $OUT/bench_gpus2_gloo.json 2> $OUT/bench_gpus2_gloo.err; echo "gpus2 rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --config 3 --steps 3 --warmup 1 > $OUT/bench_gpus2_gloo_config3.json 2> $OUT/bench_gpus2_gloo_config3.err; echo "gpus2 c3 rc=$?"
export SMMC_SEED=1 SMMC_VERBOSE=1
B=stock_market_monte_carlo_amd/bin
for PIN in 0 whole chunk; do
  for i in 1 2; do
    echo "== SMMC_PIN_HOST=$PIN benchmark_mc_gpu 1 360 100000000 (run $i)"
    SMMC_PIN_HOST=$PIN timeout -k 10 120 $B/benchmark_mc_gpu 1 360 100000000 2>&1 | grep -v "^$" | tail -6
  done
done > $OUT/cli_pin.txt 2>&1
cat $OUT/cli_pin.txt | tail -40
echo "== SMMC_DEVICE_MAP=0,0 benchmark_mc_gpu 2 360 100000000" >> $OUT/cli_pin.txt
SMMC_DEVICE_MAP=0,0 timeout -k 10 120 $B/benchmark_mc_gpu 2 360 100000000 2>&1 | tail -6 >> $OUT/cli_pin.txt
for f in $OUT/bench_*.json; do echo "$f: $(cut -c1-400 $f)"; done
