#!/bin/bash
# rocprofv3 kernel-trace summaries + plain timings of the HBM-bound "next row" kernels
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_next
mkdir -p $OUT
export TMPDIR=/tmp
python3 $R/tools/bench_stats.py > $OUT/bench_stats.jsonl 2>/dev/null || exit 1
python3 $R/tools/bench_keepdata.py > $OUT/bench_keepdata.jsonl 2>/dev/null || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_stats -- python3 $R/tools/bench_stats.py > $OUT/trace_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_keepdata -- python3 $R/tools/bench_keepdata.py > $OUT/trace_keepdata.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rd_stats -- python3 $R/tools/bench_stats.py 200000000 > $OUT/pmc_rd_stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_wr_keepdata -- python3 $R/tools/bench_keepdata.py > $OUT/pmc_wr_keepdata.log 2>&1 || exit 1
cd $R
cat $OUT/bench_stats.jsonl $OUT/bench_keepdata.jsonl
