"""PCIe-inclusive rate of smmc_engine_simulate_keepdata_to_host (trajectories into pageable host memory)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stock_market_monte_carlo_amd as S
table = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
e = S.Engine(0); e.set_table(table)
for n, p in ((1_000_000, 360), (4_000_000, 360)):
    sim = S.Engine.make_sim(n, p, S.MODE_TABLE, 7)
    e.simulate_keepdata_to_host(S.Engine.make_sim(1000, p, S.MODE_TABLE, 7))
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); traj, fin = e.simulate_keepdata_to_host(sim); dt = time.perf_counter() - t0
        best = min(best, dt)
    b = 4.0 * n * (p + 1)
    print(json.dumps({"n_paths": n, "n_periods": p, "seconds": round(best, 4), "GBps_into_host": round(b / best / 1e9, 2), "paths_per_s": n / best}))
