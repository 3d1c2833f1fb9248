"""keepdata against the number of rows: is the lower rate of long rows (P = 1000) at the usual bench size
(1.5e6 rows = 6 GB) the chunk granularity of the launch (7 chunks of 64 rows per wave) or the row length?"""
import ctypes, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stock_market_monte_carlo_amd as S
table = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
e = S.Engine(0); e.set_table(table)
for mode, name in ((S.MODE_TABLE, "table"), (S.MODE_GAUSSIAN, "gaussian")):
    for n, p in ((1_500_000, 1000), (3_000_000, 1000), (6_000_000, 1000), (12_000_000, 1000), (4_000_000, 360), (16_000_000, 360), (32_000_000, 360)):
        sim = S.Engine.make_sim(n, p, mode, 7)
        traj, fin = e.simulate_keepdata(sim, want_final=False); e.sync()
        launch = lambda: e._L.smmc_engine_simulate_keepdata(e._h, ctypes.byref(sim), ctypes.c_void_p(traj.data_ptr()), None)
        for _ in range(5): launch()
        e.sync()
        rounds = []
        for _ in range(5):
            e.timing(True)
            for _ in range(5): launch()
            ms, k = e.kernel_ms(); e.timing(False)
            rounds.append(ms / k)
        t = statistics.median(rounds)
        b = 4.0 * n * (p + 1)
        print(json.dumps({"mode": name, "n_paths": n, "n_periods": p, "GB": round(b / 1e9, 2), "kernel_ms": round(t, 4),
                          "GBps": round(b / (t * 1e-3) / 1e9, 1), "frac_of_8TBps": round(b / (t * 1e-3) / 8e12, 4)}), flush=True)
        del traj, fin
