#!/bin/bash
# Round-1 profiling recipe (run on the GPU box from the repo root through gpurun).
# Kernel-trace/stats and each PMC group are separate rocprofv3 runs.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${PROF_TAG:-prof_r01}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for MODE in gaussian table; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$MODE -- python3 $R/bench.py --mode $MODE --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace_$MODE.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq_$MODE -- python3 $R/bench.py --mode $MODE --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq_$MODE.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wr_$MODE -- python3 $R/bench.py --mode $MODE --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_wr_$MODE.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rd_$MODE -- python3 $R/bench.py --mode $MODE --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_rd_$MODE.log 2>&1 || exit 1
done
cd $R
python3 bench.py --mode table --steps 5 --warmup 1 > $OUT/bench_table.json 2> $OUT/bench_table.err || exit 1
python3 bench.py --mode gaussian --steps 10 --warmup 2 > $OUT/bench_gaussian.json 2> $OUT/bench_gaussian.err || exit 1
find $OUT -name "*.csv" | head -50
