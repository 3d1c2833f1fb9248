#!/bin/bash
# Round 4, GPU pass A: the new parity tests (Gaussian headline vs the
# reference-style CPU path; configs[3] / [4] at full size), the refreshed issue-rate tables (tools/ubench.hip,
# tools/ubench_mix.hip), reference-stream keepdata timed and its WRITE_SIZE counted, the 1e6-path step traced.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r04a
mkdir -p $OUT
cd $R
timeout -k 10 900 python -m pytest tests/test_gaussian_reference_gpu.py tests/test_full_size_configs_gpu.py tests/test_finalize_gpu.py -m gpu -q -s > $OUT/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $OUT/pytest_new.log
tail -15 $OUT/pytest_new.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --deselect tests/test_gaussian_reference_gpu.py --deselect tests/test_full_size_configs_gpu.py --deselect tests/test_finalize_gpu.py > $OUT/pytest_gpu.log 2>&1; echo "pytest all rc=$?" | tee -a $OUT/pytest_gpu.log
tail -8 $OUT/pytest_gpu.log
hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o $OUT/ubench 2> $OUT/ubench_build.log && timeout -k 10 200 $OUT/ubench > $OUT/ubench_instruction_rates.txt 2>&1; echo "ubench rc=$?"
hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o $OUT/ubench_mix 2> $OUT/ubench_mix_build.log && timeout -k 10 200 $OUT/ubench_mix > $OUT/ubench_mix.txt 2>&1; echo "ubench_mix rc=$?"
cat $OUT/ubench_mix.txt
timeout -k 10 300 python tools/bench_ref.py --traj > $OUT/bench_ref_traj.jsonl 2> $OUT/bench_ref_traj.err; echo "bench_ref traj rc=$?"
cut -c1-260 $OUT/bench_ref_traj.jsonl
export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_reftraj -- python3 $R/tools/bench_ref.py --traj > $OUT/trace_reftraj.log 2>&1; echo "trace reftraj rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wr_reftraj -- python3 $R/tools/bench_ref.py --traj > $OUT/pmc_wr_reftraj.log 2>&1; echo "pmc reftraj rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c0 -- python3 $R/bench.py --config 0 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/trace_c0.log 2>&1; echo "trace c0 rc=$?"
cd $R
F=$(find $OUT/trace_reftraj -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats_reftraj.csv
F=$(find $OUT/trace_c0 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats_c0.csv
python3 tools/pmc_summary.py $OUT/pmc_wr_reftraj > $OUT/pmc_summary_reftraj.txt 2>&1
head -40 $OUT/pmc_summary_reftraj.txt
cat $OUT/kernel_stats_c0.csv | head -8
# keep the big trace directories out of what is merged back
rm -rf $OUT/trace_reftraj $OUT/trace_c0 $OUT/ubench $OUT/ubench_mix
ls $OUT
