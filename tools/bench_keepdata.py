"""Times the keepdata kernel (every trajectory to HBM, path-major): the one HBM-bound kernel."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stock_market_monte_carlo_amd as S
table = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
e = S.Engine(0); e.set_table(table)
out = []
for mode, name in ((S.MODE_TABLE, "table"), (S.MODE_GAUSSIAN, "gaussian")):
    for n, p in ((4_000_000, 360), (1_500_000, 1000), (16_000_000, 63)):
        sim = S.Engine.make_sim(n, p, mode, 7)
        traj, fin = e.simulate_keepdata(sim); e.sync()
        e.timing(True)
        for _ in range(20):  # short kernels: the first few launches after idle run slower
            e._L.smmc_engine_simulate_keepdata(e._h, __import__("ctypes").byref(sim), __import__("ctypes").c_void_p(traj.data_ptr()), None)
        ms, k = e.kernel_ms(); e.timing(False)
        b = 4.0 * n * (p + 1)
        out.append({"mode": name, "n_paths": n, "n_periods": p, "kernel_ms": ms / k, "GBps": b / (ms / k * 1e-3) / 1e9,
                    "paths_per_s": n / (ms / k * 1e-3), "frac_of_8TBps": b / (ms / k * 1e-3) / 8e12})
        del traj, fin
for o in out: print(json.dumps(o))
