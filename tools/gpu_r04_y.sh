#!/bin/bash
# Round 4, GPU pass Y: the three bench lines that quote the profile pass (traffic, weighted_frac), on the final library.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r04y; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
for c in 1 2; do timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_config${c}_after_profile.json 2>/dev/null; python -c "import sys,json; d=json.loads(open('$O/bench_config${c}_after_profile.json').read()); print('config$c', '%.4g' % d['value'], d['roofline']['traffic'], d['valu']['weighted_frac'], d['valu']['held_clock_ghz'], d['hbm_bound_kernels']['keepdata']['kernel_ms'])"; done
timeout -k 10 300 python bench.py --stream ref --outputs final --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_stream_ref_after_profile.json 2>/dev/null; echo "ref rc=$?"
