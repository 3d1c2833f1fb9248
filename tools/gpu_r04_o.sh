#!/bin/bash
# Round 4, GPU pass O: board power and clocks under keepdata, paths_kernel and a plain fill (tools/power_probe.py).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r04o; mkdir -p $O
python -c "from stock_market_monte_carlo_amd import build; import sys; sys.exit(1 if build.stale() else 0)" || { echo "library is stale in the snapshot: rebuilding on the box"; python -m stock_market_monte_carlo_amd.build > /dev/null 2>&1; }
rocm-smi --showpower --showclocks --showtemp --json > $O/smi_once.json 2>$O/smi_once.err; echo "rocm-smi rc=$?"; head -c 1500 $O/smi_once.json; echo
rocm-smi --showmaxpower --showperflevel --showclkfrq 2>&1 | head -60 > $O/smi_caps.txt
timeout -k 10 400 python tools/power_probe.py > $O/power_probe.jsonl 2>$O/power_probe.err; echo "probe rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r04o/power_probe.jsonl"):
    j = json.loads(l)
    print(j["case"], j["child"])
    for s in j["samples"][-3:]:
        print("   ", s)
PY
tail -5 $O/power_probe.err
