"""Development: keepdata and paths kernel times against problem size (clock ramp / launch effects)."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stock_market_monte_carlo_amd as S
table = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
e = S.Engine(0); e.set_table(table)
reps = int(os.environ.get("REPS", "20"))
for n in (1_000_000, 4_000_000, 16_000_000, 32_000_000):
    sim = S.Engine.make_sim(n, 360, S.MODE_TABLE, 7)
    traj = torch.empty(n * 361, dtype=torch.float32, device="cuda")
    fin = torch.empty(n, dtype=torch.float32, device="cuda")
    for name, call in (("keepdata", lambda: e._L.smmc_engine_simulate_keepdata(e._h, ctypes.byref(sim), ctypes.c_void_p(traj.data_ptr()), None)),
                       ("paths", lambda: e._L.smmc_engine_simulate(e._h, ctypes.byref(sim), ctypes.c_void_p(fin.data_ptr()), None, None, None))):
        call(); e.sync()
        e.timing(True)
        for _ in range(reps): call()
        ms, k = e.kernel_ms(); e.timing(False)
        t = ms / k
        print(json.dumps({"kernel": name, "n_paths": n, "ms": round(t, 4), "periods_per_s": n * 360 / (t * 1e-3),
                          "GBps": (4.0 * n * 361 / (t * 1e-3) / 1e9) if name == "keepdata" else None}))
    del traj, fin
