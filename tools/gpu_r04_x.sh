#!/bin/bash
# (Record of an experiment: SMMC_KEEPDATA_REST existed only in the build these runs measured; the form was not kept -- DESIGN.md section 5.)
# Round 4, GPU pass X: keepdata inside bench.py (after 20 steps of paths_kernel, the statistics and the quartiles) with the rest
# rows beside the comb kernel and after it.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r04x; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
for round in 1 2 3; do
  for form in beside serial; do
    SMMC_KEEPDATA_REST=$form timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['hbm_bound_kernels']; print('$form', 'keepdata ms %.4f' % h['keepdata']['kernel_ms'], 'vs fill %.3f' % h['keepdata']['vs_box_fill'], 'value %.4g' % d['value'])"
  done
done | tee $O/bench_keepdata_rest_forms.txt
