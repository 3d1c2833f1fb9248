#!/bin/bash
# Round 4, GPU pass K: keepdata against bare store loops ON ONE BOX (VERDICT r3 item 8 asks for the counters / micro-benchmarks
# before any kernel change): tools/ubench_store_pattern.hip (contiguous fill, comb pattern, any grid), tools/ubench_comb_stores.hip
# (the kernel's pattern in the kernel's launch shape), tools/bench_keepdata.py, twice interleaved; then the bench line.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04k; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
hipcc -O3 --offload-arch=gfx950 tools/ubench_store_pattern.hip -o /tmp/ubench_store_pattern || exit 1
hipcc -O3 --offload-arch=gfx950 tools/ubench_comb_stores.hip -o /tmp/ubench_comb_stores || exit 1
for round in 1 2; do
  timeout -k 10 120 /tmp/ubench_store_pattern 361 2>&1 | grep -v "C unaligned" > $O/store_pattern_$round.txt || exit 1
  timeout -k 10 200 /tmp/ubench_comb_stores > $O/comb_stores_$round.txt 2>&1 || exit 1
  timeout -k 10 200 python tools/bench_keepdata.py > $O/bench_keepdata_$round.jsonl 2>$O/bench_keepdata_$round.err || exit 1
  grep "B contiguous\|D comb, 1 rows" $O/store_pattern_$round.txt
  grep "one workgroup per CU" $O/comb_stores_$round.txt | grep "12 waves\|16 waves"
  python -c "
import json
for l in open('$O/bench_keepdata_$round.jsonl'):
    j = json.loads(l); print(j['mode'], j['n_paths'], j['n_periods'], j['kernel_ms'], j['GBps'])"
done
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > $O/bench_config1.json 2>$O/bench_config1.err; echo "bench rc=$?"
python -c "
import json
j = json.loads([l for l in open('$O/bench_config1.json') if l.startswith('{\"metric\"')][-1])
print(j['value'], j['ms_per_step'], j['valu']['weighted_frac'], j['valu']['weighted_frac_measured_costs'], j['valu']['held_clock_ghz'])
print(json.dumps(j['hbm_bound_kernels'], indent=1))"
