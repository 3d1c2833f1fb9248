#!/bin/bash
# bench.py over 5000 steps (about 70 s of back-to-back launches) beside the driver-length run on the same box:
# the rate the chip holds once clocks and temperature have settled.  Output: gpurun_out/r03s/bench_*.json
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03s
mkdir -p $OUT
cd $R
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_short_config1.json 2> /dev/null || exit 1
timeout -k 10 300 python bench.py --steps 5000 --warmup 50 --no-cpu-baseline > $OUT/bench_sustained_config1.json 2> /dev/null || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_short_after_config1.json 2> /dev/null || exit 1
timeout -k 10 300 python bench.py --config 2 --steps 5000 --warmup 50 --no-cpu-baseline > $OUT/bench_sustained_config2.json 2> /dev/null || exit 1
timeout -k 10 300 python bench.py --stream ref --outputs final --steps 2000 --warmup 20 --no-cpu-baseline > $OUT/bench_sustained_stream_ref.json 2> /dev/null || exit 1
python - <<'PY'
import json
for f in ("short_config1", "sustained_config1", "short_after_config1", "sustained_config2", "sustained_stream_ref"):
    d = json.loads([l for l in open("gpurun_out/r03s/bench_%s.json" % f) if l.startswith("{")][0])
    print(f, "steps", d["steps"], "paths/s %.4g" % d["value"], "ms/step %.3f" % d["ms_per_step"], "kernel_ms %.3f" % d["roofline"]["kernel_ms"], "valu %.3f" % d["valu"]["frac"])
PY
