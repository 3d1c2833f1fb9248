#!/bin/bash
# Development: bench.py configs 1 and 2 against several builds of the library, interleaved, on one box:
#   tools/gpu_ab_cfg.sh OUT_TAG LIB...     ("product" = the in-tree library)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT
cd $R
for ROUND in 1 2; do
  for L in "$@"; do
    P=$L; [ $L = product ] && P=$R/stock_market_monte_carlo_amd/libsmmc_hip.so
    for CFG in 1 2; do
      SMMC_LIB=$P timeout -k 10 200 python bench.py --config $CFG --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L config$CFG', '%.4g' % d['value'], '%.3f' % d['roofline']['kernel_ms'])" || exit 1
    done
  done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
