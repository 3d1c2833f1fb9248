"""simulate_to_host of 1e8 x 360 table paths into pinned host memory: time against the chunk length
(SMMC_HOST_CHUNK_PATHS, one fresh engine each) and with / without a progress poller.  One JSON line per case."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import stock_market_monte_carlo_amd as S  # noqa: E402

table = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
n = 100_000_000
host = torch.empty(n, dtype=torch.float32).pin_memory().numpy()
for chunk in (1 << 24, 1 << 23, 6250496, 1 << 22, 1 << 21):
    os.environ["SMMC_HOST_CHUNK_PATHS"] = str(chunk)
    e = S.Engine(0, stream="new")
    e.set_table(table)
    sim = S.Engine.make_sim(n, 360, S.MODE_TABLE, 11)
    for polled in (False, True):
        prog = C.c_int64(0) if polled else None
        e.simulate_to_host(sim, out=host, progress=prog)
        t0 = time.perf_counter()
        for _ in range(5):
            e.simulate_to_host(sim, out=host, progress=prog)
        dt = (time.perf_counter() - t0) / 5
        print(json.dumps({"chunk_paths": chunk, "chunks": -(-n // chunk), "polled": polled, "ms": dt * 1e3,
                          "GBps_to_host": 4.0 * n / dt / 1e9}), flush=True)
    e.close()
