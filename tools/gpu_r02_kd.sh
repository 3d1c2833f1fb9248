#!/bin/bash
# Round-2 keepdata pass: counter passes over the comb kernel (both modes), the interleaved A/B of its
# launch knobs, and the resident-values statistics kernels.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
mkdir -p gpurun_out/r02kd
TAG=g_comb KD_MODE=gaussian bash tools/kd_pmc.sh > gpurun_out/r02kd/pmc_g.txt 2>&1 || exit 1
TAG=t_comb KD_MODE=table bash tools/kd_pmc.sh > gpurun_out/r02kd/pmc_t.txt 2>&1 || exit 1
KD_MODES=gaussian KD_SHAPES=4000000x360,1500000x1000 timeout -k 10 300 python tools/kd_ab.py "SMMC_KEEPDATA_K=2" "SMMC_KEEPDATA_K=1" "SMMC_KEEPDATA_K=4" "SMMC_KEEPDATA_K=2,SMMC_KEEPDATA_COMB_WAVES=12" "SMMC_KEEPDATA_K=2,SMMC_KEEPDATA_COMB_ILP=1" "SMMC_KEEPDATA_KERNEL=tile" "" > gpurun_out/r02kd/kd_ab.txt 2>&1 || exit 1
KD_MODES=table KD_SHAPES=4000000x360,1500000x1000 timeout -k 10 300 python tools/kd_ab.py "SMMC_KEEPDATA_K=1" "SMMC_KEEPDATA_K=2" "SMMC_KEEPDATA_KERNEL=tile" "" >> gpurun_out/r02kd/kd_ab.txt 2>&1 || exit 1
cat gpurun_out/r02kd/kd_ab.txt
timeout -k 10 200 python tools/bench_stats.py 100000000 > gpurun_out/r02kd/bench_stats.jsonl 2>/dev/null || exit 1
timeout -k 10 200 python tools/bench_stats.py 1000000000 >> gpurun_out/r02kd/bench_stats.jsonl 2>/dev/null || exit 1
cut -c1-200 gpurun_out/r02kd/bench_stats.jsonl
