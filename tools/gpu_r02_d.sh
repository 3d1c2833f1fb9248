#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r02d
mkdir -p $OUT
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
grep -q " passed" $OUT/pytest_gpu.log || exit 1
for c in 1 2; do timeout -k 10 300 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_config$c.json 2> $OUT/bench_config$c.err; echo "config$c rc=$?"; done
python - <<'PY'
import json
for c in (1,2):
    d=json.loads([l for l in open(f"gpurun_out/r02d/bench_config{c}.json") if l.startswith("{")][0])
    print(c, "%.4g paths/s"%d["value"], "kernel_ms %.3f"%d["roofline"]["kernel_ms"], "valu frac %.3f"%d["valu"]["frac"], {k:(round(v["kernel_ms"],3), round(v["frac_of_peak"],3)) for k,v in d["hbm_bound_kernels"].items()})
PY
ROUNDS=4 KD_SHAPES=4000000x360,1500000x1000 timeout -k 10 600 python tools/kd_ab.py "SMMC_KEEPDATA_KERNEL=tile" "SMMC_KEEPDATA_KERNEL=comb,SMMC_KEEPDATA_K=1,SMMC_KEEPDATA_COMB_ILP=1" "SMMC_KEEPDATA_KERNEL=comb,SMMC_KEEPDATA_K=1" "SMMC_KEEPDATA_KERNEL=comb,SMMC_KEEPDATA_K=2" "SMMC_KEEPDATA_KERNEL=comb,SMMC_KEEPDATA_K=1,SMMC_KEEPDATA_COMB_WAVES=12" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab.txt
