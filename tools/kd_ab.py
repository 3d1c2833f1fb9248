"""Development: interleaved A/B of keepdata variants (environment knobs are read per call).
usage: kd_ab.py "K=V,K=V" "K=V" ...   (each argument one variant; KD_MODE / KD_N / KD_P select the workload)"""
import ctypes, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import stock_market_monte_carlo_amd as S
table = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
variants = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
knobs = sorted({k for v in variants for k in v})
e = S.Engine(0); e.set_table(table)
for mode_name in os.environ.get("KD_MODES", "table,gaussian").split(","):
    mode = S.MODE_GAUSSIAN if mode_name == "gaussian" else S.MODE_TABLE
    shapes = [tuple(int(x) for x in sh.split("x")) for sh in os.environ.get("KD_SHAPES", "4000000x360,1500000x1000,16000000x63").split(",")]
    for n, p in shapes:
        sim = S.Engine.make_sim(n, p, mode, 7)
        traj = torch.empty(n * (p + 1), dtype=torch.float32, device="cuda")
        times = [[] for _ in variants]
        for rnd in range(int(os.environ.get("ROUNDS", "6"))):
            for vi, v in enumerate(variants):
                for kname in knobs:
                    os.environ.pop(kname, None)
                os.environ.update(v)
                e._L.smmc_engine_simulate_keepdata(e._h, ctypes.byref(sim), ctypes.c_void_p(traj.data_ptr()), None)
                e.sync(); e.timing(True)
                for _ in range(8):
                    e._L.smmc_engine_simulate_keepdata(e._h, ctypes.byref(sim), ctypes.c_void_p(traj.data_ptr()), None)
                ms, k = e.kernel_ms(); e.timing(False)
                if rnd: times[vi].append(ms / k)
        b = 4.0 * n * (p + 1)
        print("%-8s P=%-5d " % (mode_name, p) + "  ".join("[%s] %.3f ms %.0f GB/s" % (",".join("%s=%s" % (k.replace("SMMC_KEEPDATA_", ""), x) for k, x in v.items()), statistics.median(t), b / (statistics.median(t) * 1e-3) / 1e9) for v, t in zip(variants, times)), flush=True)
        del traj
