#!/bin/bash
# (Record of an experiment: SMMC_KEEPDATA_REST existed only in the build these runs measured; the form was not kept -- DESIGN.md section 5.)
# Round 4, GPU pass U: keepdata with the rest rows beside the comb kernel (product) against after it (SMMC_KEEPDATA_REST=serial),
# interleaved; before that the keepdata test files.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=$R/gpurun_out/r04u; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
timeout -k 10 900 python -m pytest tests/test_keepdata_comb_gpu.py tests/test_gpu_parity.py tests/test_fuzz_gpu.py tests/test_dropin_gpu.py tests/test_host_pipeline_gpu.py -m gpu -q -x > $O/pytest_keepdata.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_keepdata.log
tail -3 $O/pytest_keepdata.log
grep -q "pytest rc=0" $O/pytest_keepdata.log || exit 1
for round in 1 2 3; do
  for form in beside serial; do
    echo "== $form"
    SMMC_KEEPDATA_REST=$form timeout -k 10 200 python tools/bench_keepdata.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); print('$form', j['mode'], j['n_paths'], j['n_periods'], j['kernel_ms'], j['GBps'])"
  done
done | tee $O/keepdata_rest_forms.txt
