#!/bin/bash
# End-to-end runs of the drop-in command lines (host buffers, PCIe included), as the
# reference's README benchmarks them: build/benchmark_mc_* 360 100000000 (README.md:80-85)
export SMMC_SEED=1
B=stock_market_monte_carlo_amd/bin
for i in 1 2; do   # second iteration is warm (engine creation, page faults of the result vector)
echo "== benchmark_mc_gpu 1 360 100000000 (table mode)"; $B/benchmark_mc_gpu 1 360 100000000 | tail -3 || exit 1
done
echo; echo "== SMMC_MODE=gaussian benchmark_mc_gpu 1 360 100000000"; SMMC_MODE=gaussian $B/benchmark_mc_gpu 1 360 100000000 | tail -3 || exit 1
echo; echo "== benchmark_mc_gpu_reduceBlock 1 360 100000000"; $B/benchmark_mc_gpu_reduceBlock 1 360 100000000 | tail -6 || exit 1
echo; echo "== benchmark_mc_cpu_v2 360 100000000"; $B/benchmark_mc_cpu_v2 360 100000000 | tail -1 || exit 1
echo; echo "== benchmark_mc_cpu 360 2000000 (keepdata: 2.9 GB of trajectories into vector<vector<float>>)"; $B/benchmark_mc_cpu 360 2000000 | tail -1 || exit 1
echo; echo "== benchmark_reduce_mean 1000000000"; $B/benchmark_reduce_mean 1000000000 | tail -4 || exit 1
