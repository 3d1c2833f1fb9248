#!/bin/bash
# GPU box: the statistics / order-statistics tests alone (a quick confirmation after a change to smmc_stats_kernels.hip).
cd ${GRAFT_REPO_ROOT:-$PWD}
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1
timeout -k 10 500 python -m pytest tests/test_stats_gpu.py -m gpu -q -x 2>&1 | tail -15
