#!/bin/bash
# Round 4, GPU pass J: the round's profile pass (tools/profile_r04.sh) -- kernel traces, PMC passes, pmc_traffic.json with the
# priced loops -- then bench.py config 1 / 2 against the fresh pmc_traffic.json (traffic and weighted_frac in the line).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
PROF_TAG=prof_r04 timeout -k 10 1000 bash tools/profile_r04.sh > gpurun_out/prof_r04.log 2>&1; echo "profile rc=$?"
tail -5 gpurun_out/prof_r04.log
OUT=$R/gpurun_out/prof_r04/keep
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
for c in 1 2; do timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_config${c}_after_profile.json 2>/dev/null; python -c "import sys,json; d=json.loads(open('$OUT/bench_config${c}_after_profile.json').read()); print('config$c', '%.4g' % d['value'], d['roofline']['traffic'], d['valu']['weighted_frac'], d['valu']['held_clock_ghz'])"; done
timeout -k 10 300 python bench.py --stream ref --outputs final --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_stream_ref_after_profile.json 2>/dev/null
head -60 $OUT/pmc_summary.txt | cut -c1-160
cat $OUT/pmc_summary_reftraj.txt | grep -A3 "ref_windowed_kernel\|ref_tree_kernel" | cut -c1-160
