#!/bin/bash
for T in 32 64; do for B in 2 4 8 16 64; do
  echo "tile=$T bpc=$B"; SMMC_KEEPDATA_TILE=$T SMMC_KEEPDATA_BLOCKS_PER_CU=$B python3 tools/bench_keepdata.py 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-8s P=%-5d %.3f ms  %.0f GB/s'%(d['mode'],d['n_periods'],d['kernel_ms'],d['GBps']))" || exit 1
done; done
