#!/bin/bash
# Round 4, GPU pass Z: the statistics tests with the order-statistics fuzz, table form and (SMMC_RADIX_MATCH=chain) chain form.
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1
timeout -k 10 600 python -m pytest tests/test_stats_gpu.py -m gpu -q -x 2>&1 | tail -5
SMMC_RADIX_MATCH=chain timeout -k 10 600 python -m pytest tests/test_stats_gpu.py -m gpu -q -x -k fuzz 2>&1 | tail -3
