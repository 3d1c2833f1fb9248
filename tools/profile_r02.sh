#!/bin/bash
# Round-2 profiling recipe (GPU box, from the repo root through gpurun).  Kernel-trace/stats and each
# PMC group are separate rocprofv3 runs of the SAME command line bench.py is judged on.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${PROF_TAG:-prof_r02}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for CFG in 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c$CFG -- python3 $R/bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace_c$CFG.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq_c$CFG -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq_c$CFG.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wr_c$CFG -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_wr_c$CFG.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rd_c$CFG -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_rd_c$CFG.log 2>&1 || exit 1
done
cd $R
python3 tools/pmc_summary.py $OUT/pmc_sq_c1 $OUT/pmc_wr_c1 $OUT/pmc_rd_c1 $OUT/pmc_sq_c2 $OUT/pmc_wr_c2 $OUT/pmc_rd_c2 > $OUT/pmc_summary.txt
python3 tools/pmc_traffic.py --key "gaussian|100000000|360|all" --write $OUT/pmc_wr_c1 --fetch $OUT/pmc_rd_c1 --source "profiles/r02/pmc_summary.txt (tools/profile_r02.sh)" --out $OUT/pmc_traffic.json
python3 tools/pmc_traffic.py --key "table|100000000|360|all" --write $OUT/pmc_wr_c2 --fetch $OUT/pmc_rd_c2 --source "profiles/r02/pmc_summary.txt (tools/profile_r02.sh)" --out $OUT/pmc_traffic.json
find $OUT -name "*kernel_stats.csv" | head
