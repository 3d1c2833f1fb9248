#!/bin/bash
# Copies what tools/gpu_r02_final.sh, tools/profile_r02.sh and tools/gpu_r02_kd.sh left under gpurun_out/
# (scratch) into profiles/r02/ (tracked).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
G=$R/gpurun_out; P=$R/profiles/r02
for f in bench_config0 bench_config1 bench_config2 bench_config3 bench_config3_share bench_config4 bench_config4_share \
         bench_gpus2_gloo bench_gpus2_gloo_config3 bench_gpus2_gloo_config4 bench_rehearse_rccl; do cp $G/r02f/$f.json $P/$f.json; done
cp $G/r02f/bench_keepdata.jsonl $G/r02f/cli_runs.txt $P/
cp $G/prof_r02f/pmc_summary.txt $P/pmc_summary.txt
cp $G/prof_r02f/pmc_traffic.json $R/profiles/pmc_traffic.json
for c in 1 2; do
  cp "$(find $G/prof_r02f/trace_c$c -name '*kernel_stats.csv' -printf '%T@ %p\n' | sort -n | tail -1 | cut -d' ' -f2)" $P/kernel_stats_config$c.csv
  grep "^{" $G/prof_r02f/trace_c$c.log > $P/bench_under_rocprof_config$c.json
done
{ echo "# rocprofv3 --pmc passes over keepdata (tools/kd_pmc.sh, tools/kd_one.py: 4e6 paths x 361 values = 5.776 GB per launch), final build"
  echo "# keepdata_kernel<...> lines: the tile kernel on the last < 2048 rows of the call"
  echo "# comb kernel, Gaussian mode (K = 1 row per stream, 2 Philox blocks per step, 14 waves per CU)"
  grep -v "^== \|copyBuffer" $G/kd_pmc_g_comb/summary.txt
  echo "# comb kernel, table mode (K = 1, 12 waves per CU)"
  grep -v "^== \|copyBuffer" $G/kd_pmc_t_comb/summary.txt; } > $P/pmc_summary_keepdata.txt
grep -v amdgpu.ids $G/r02kd/kd_ab.txt > $P/keepdata_ab_stream_v3.txt
cp $G/r02kd/bench_stats.jsonl $P/bench_stats.jsonl
echo collected
