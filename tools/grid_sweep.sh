#!/bin/bash
# bench.py configs[1] / configs[2] against the paths kernel's workgroups per CU (SMMC_BLOCKS_PER_CU; default 64), three
# rounds interleaved on one box.  Output: one line per run "<config> <blocks per CU> <kernel ms> <paths/s>".
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for round in 1 2 3; do
  for c in 1 2; do
    for b in 16 32 64 128 256 512 1536; do
      SMMC_BLOCKS_PER_CU=$b timeout -k 10 120 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config', $c, 'blocks_per_cu', $b, 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'paths_per_s %.4g' % d['value'])" || exit 1
    done
  done
done
