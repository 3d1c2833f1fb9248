"""PCIe-inclusive rate: final values of 1e8 paths delivered into HOST memory through the
chunked compute/copy pipeline (smmc_engine_simulate_to_host).  Not the headline `value`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import stock_market_monte_carlo_amd as S
table = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
e = S.Engine(0, stream="new"); e.set_table(table)
n = 100_000_000
for mode, name in ((S.MODE_TABLE, "table"), (S.MODE_GAUSSIAN, "gaussian")):
    sim = S.Engine.make_sim(n, 360, mode, 11)
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            t = torch.empty(n, dtype=torch.float32).pin_memory(); host = t.numpy()
        else:
            host = np.empty(n, dtype=np.float32); host[:] = 0  # touch the pages
        e.simulate_to_host(sim, out=host)  # warm-up
        t0 = time.perf_counter()
        for _ in range(3):
            e.simulate_to_host(sim, out=host)
        dt = (time.perf_counter() - t0) / 3
        print(json.dumps({"mode": name, "host_memory": kind, "n_paths": n, "seconds": dt, "paths_per_s": n / dt,
                          "GBps_to_host": 4.0 * n / dt / 1e9}))
