for B in 2 4 8 16 32; do for N in 100000000 1000000000; do echo -n "bpc=$B n=$N: "; SMMC_STATS_BLOCKS_PER_CU=$B python3 tools/bench_stats.py $N 2>/dev/null | head -2 | python3 -c "
import sys,json
print(' | '.join('%s %.3f ms %.0f GB/s'%(d['kernel'][:22],d['ms_per_launch'],d['GBps']) for d in map(json.loads, sys.stdin)))"; done; done
