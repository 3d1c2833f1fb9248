#!/bin/bash
# Round 4, GPU pass L: which store shapes the memory system prefers, on one box: the comb pattern with 1 / 2 / 4 / 8 consecutive
# lines per stream per visit, plain 16-byte fills (grid-stride, one piece per workgroup, hipMemsetAsync), and the kernel.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04l; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
hipcc -O3 --offload-arch=gfx950 tools/ubench_store_pattern.hip -o /tmp/ubench_store_pattern || exit 1
for round in 1 2; do
  timeout -k 10 120 /tmp/ubench_store_pattern 361 skip > $O/store_pattern_$round.txt 2>&1 || exit 1
  grep "B contiguous\|D comb, 1 rows\|E comb\|H comb\|fill\|Memset" $O/store_pattern_$round.txt
done
timeout -k 10 200 python tools/bench_keepdata.py > $O/bench_keepdata.jsonl 2>$O/bench_keepdata.err || exit 1
python -c "
import json
for l in open('$O/bench_keepdata.jsonl'):
    j = json.loads(l); print(j['mode'], j['n_paths'], j['n_periods'], j['kernel_ms'], j['GBps'])"
