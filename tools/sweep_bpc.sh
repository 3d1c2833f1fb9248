#!/bin/bash
# blocks-per-CU sweep of the paths kernel (development)
for M in table gaussian; do for B in ${BPCS:-4 5 6 7 8 12 16}; do
  echo -n "$M bpc=$B: "; SMMC_BLOCKS_PER_CU=$B python3 bench.py --mode $M --steps 6 --warmup 2 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4g paths/s  kernel_ms=%.3f'%(d['value'], d['roofline']['kernel_ms']))" || exit 1
done; done
