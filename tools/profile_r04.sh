#!/bin/bash
# Round-4 profiling recipe (GPU box, from the repo root through gpurun).  Kernel-trace/stats and each
# PMC group are separate rocprofv3 runs of the SAME command line bench.py is judged on: configs[1]
# (Gaussian, the headline), configs[2] (table) and the reference CPU stream on the device (--stream ref: 360
# periods = ref_windowed_kernel, 1000 periods = ref_tree_kernel).
# Writes gpurun_out/$PROF_TAG/{trace_*,pmc_*}, pmc_summary.txt and pmc_traffic.json (with the ISA
# fingerprint and source digest of the build that was profiled); copy what is judged into profiles/r04/
# and pmc_traffic.json to profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${PROF_TAG:-prof_r04}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run_passes() {  # tag, bench arguments...
  local TAG=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -- python3 $R/bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace_$TAG.log 2>&1 || return 1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq_$TAG -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq_$TAG.log 2>&1 || return 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wr_$TAG -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_wr_$TAG.log 2>&1 || return 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rd_$TAG -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_rd_$TAG.log 2>&1 || return 1
  echo "passes done: $TAG"
}
run_passes c1 --config 1 || exit 1
run_passes c2 --config 2 || exit 1
run_passes ref --config 2 --stream ref --outputs final || exit 1
run_passes ref1000 --config 2 --stream ref --outputs final --periods 1000 --paths-per-gpu 20000000 || exit 1
# reference-stream trajectories: kernel trace + WRITE_SIZE of tools/bench_ref.py --traj
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_reftraj -- python3 $R/tools/bench_ref.py --traj > $OUT/trace_reftraj.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wr_reftraj -- python3 $R/tools/bench_ref.py --traj > $OUT/pmc_wr_reftraj.log 2>&1 || exit 1
# the 1e6-path step (BASELINE configs[0] size): one launch per step now
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c0 -- python3 $R/bench.py --config 0 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/trace_c0.log 2>&1 || exit 1
cd $R
python3 tools/pmc_summary.py $OUT/pmc_wr_reftraj > $OUT/pmc_summary_reftraj.txt
F=$(find $OUT/trace_reftraj -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats_reftraj.csv
F=$(find $OUT/trace_c0 -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/kernel_stats_c0.csv
grep '^{"metric"' $OUT/trace_c0.log | tail -1 > $OUT/bench_under_rocprof_c0.json
python3 tools/pmc_summary.py $OUT/pmc_sq_c1 $OUT/pmc_wr_c1 $OUT/pmc_rd_c1 $OUT/pmc_sq_c2 $OUT/pmc_wr_c2 $OUT/pmc_rd_c2 \
        $OUT/pmc_sq_ref $OUT/pmc_wr_ref $OUT/pmc_rd_ref $OUT/pmc_sq_ref1000 $OUT/pmc_wr_ref1000 $OUT/pmc_rd_ref1000 > $OUT/pmc_summary.txt
SRC="profiles/r04/pmc_summary.txt (tools/profile_r04.sh)"
python3 tools/pmc_traffic.py --key "gaussian|100000000|360|all" --write $OUT/pmc_wr_c1 --fetch $OUT/pmc_rd_c1 --source "$SRC" --out $OUT/pmc_traffic.json
python3 tools/pmc_traffic.py --key "table|100000000|360|all" --write $OUT/pmc_wr_c2 --fetch $OUT/pmc_rd_c2 --source "$SRC" --out $OUT/pmc_traffic.json
python3 tools/pmc_traffic.py --key "ref|100000000|360|final" --write $OUT/pmc_wr_ref --fetch $OUT/pmc_rd_ref --source "$SRC" --out $OUT/pmc_traffic.json
python3 tools/pmc_traffic.py --key "ref|20000000|1000|final" --write $OUT/pmc_wr_ref1000 --fetch $OUT/pmc_rd_ref1000 --source "$SRC" --out $OUT/pmc_traffic.json
for T in c1 c2 ref ref1000; do
  F=$(find $OUT/trace_$T -name "*kernel_stats.csv" | head -1)
  [ -n "$F" ] && cp $F $OUT/kernel_stats_$T.csv
  grep '^{"metric"' $OUT/trace_$T.log | tail -1 > $OUT/bench_under_rocprof_$T.json
done
# what is judged: small summaries only (the trace directories stay on the box)
mkdir -p $OUT/keep && cp $OUT/*.txt $OUT/*.csv $OUT/*.json $OUT/keep/ 2>/dev/null
rm -rf $OUT/trace_* $OUT/pmc_sq_* $OUT/pmc_wr_* $OUT/pmc_rd_*
ls $OUT/keep
