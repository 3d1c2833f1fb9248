"""Development: one keepdata configuration, repeated (for rocprofv3 counter passes)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stock_market_monte_carlo_amd as S
table = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
mode = S.MODE_GAUSSIAN if os.environ.get("KD_MODE") == "gaussian" else S.MODE_TABLE
n, p, reps = int(os.environ.get("KD_N", "4000000")), int(os.environ.get("KD_P", "360")), int(os.environ.get("REPS", "6"))
e = S.Engine(0); e.set_table(table)
sim = S.Engine.make_sim(n, p, mode, 7)
traj = torch.empty(n * (p + 1), dtype=torch.float32, device="cuda")
e.timing(True)
for _ in range(reps):
    e._L.smmc_engine_simulate_keepdata(e._h, ctypes.byref(sim), ctypes.c_void_p(traj.data_ptr()), None)
ms, k = e.kernel_ms(); e.timing(False)
print("keepdata n=%d p=%d: %.4f ms  %.0f GB/s" % (n, p, ms / k, 4.0 * n * (p + 1) / (ms / k * 1e-3) / 1e9))
