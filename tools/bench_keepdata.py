"""Times the keepdata kernel (every trajectory to HBM, path-major): the one HBM-bound kernel.
Per configuration: 10 untimed launches, then 5 rounds of 10 timed launches (HIP events around each
launch, on the engine's stream); reports the median round."""
import ctypes, json, os, statistics, sys
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
        launch = lambda: e._L.smmc_engine_simulate_keepdata(e._h, ctypes.byref(sim), ctypes.c_void_p(traj.data_ptr()), None)
        for _ in range(10): launch()
        e.sync()
        rounds = []
        for _ in range(5):
            e.timing(True)
            for _ in range(10): launch()
            ms, k = e.kernel_ms(); e.timing(False)
            rounds.append(ms / k)
        t = statistics.median(rounds)
        b = 4.0 * n * (p + 1)
        out.append({"mode": name, "n_paths": n, "n_periods": p, "kernel_ms": round(t, 4), "GBps": round(b / (t * 1e-3) / 1e9, 1),
                    "paths_per_s": n / (t * 1e-3), "frac_of_8TBps": round(b / (t * 1e-3) / 8e12, 4),
                    "rounds_ms": [round(r, 4) for r in rounds]})
        del traj, fin
for o in out: print(json.dumps(o))
