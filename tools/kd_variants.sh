#!/bin/bash
# keepdata kernel variants (development): parity tests, then timings, for each "tile waves" pair
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/kd_variants
mkdir -p $OUT
IFS=";" read -ra VARIANTS <<< "${KD_VARIANTS:-32 4;16 4;32 8}"
for V in "${VARIANTS[@]}"; do
  set -- $V
  export SMMC_KEEPDATA_TILE=$1
  if [ -n "$2" ]; then export SMMC_KEEPDATA_WAVES=$2; else unset SMMC_KEEPDATA_WAVES; fi
  echo "== tile $1 waves ${2:-default}"
  timeout -k 10 300 python3 -m pytest $R/tests/test_gpu_parity.py $R/tests/test_dropin_gpu.py -m gpu -x -q -k "keepdata or dropin" > $OUT/tests_$1_$2.log 2>&1 || { tail -20 $OUT/tests_$1_$2.log; exit 1; }
  tail -1 $OUT/tests_$1_$2.log
  timeout -k 10 200 python3 $R/tools/bench_keepdata.py 2>/dev/null | tee $OUT/bench_$1_$2.jsonl | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-8s P=%-5d %.3f ms  %.0f GB/s'%(d['mode'],d['n_periods'],d['kernel_ms'],d['GBps']))" || exit 1
done
