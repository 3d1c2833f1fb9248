"""Times the HBM-read-bound statistics kernels on device-resident final values."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stock_market_monte_carlo_amd as S
e = S.Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
r = e.simulate(S.Engine.make_sim(n, 36, S.MODE_GAUSSIAN, 3))
e.sync()
v = r.final
for name, fn, passes in (("values_stats(100 bins)", lambda: e.values_stats(v, 1000.0, 100, 0.0, 20000.0), 1),
                         ("values_stats(no hist)", lambda: e.values_stats(v, 1000.0), 1),
                         ("quartiles(3 radix passes)", lambda: e.quartiles(v), 3)):
    fn(); e.sync()
    e.timing(True)
    for _ in range(5):
        fn()
    ms, k = e.kernel_ms(); e.timing(False)
    per = ms / k
    print(json.dumps({"kernel": name, "n": n, "launches": k, "ms_per_launch": per, "GBps": 4.0 * n / (per * 1e-3) / 1e9,
                      "frac_of_8TBps": 4.0 * n / (per * 1e-3) / 8e12, "ms_per_call": ms / 5}))
