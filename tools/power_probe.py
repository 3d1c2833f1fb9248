"""Development: board power and clocks (rocm-smi, sampled from a parent that never touches the GPU) while a child
runs one kernel back to back for a few seconds: keepdata (table / Gaussian), paths_kernel (Gaussian / table), a
plain fill.  Answers whether a kernel that is bound by neither its VALU work, its LDS cycles nor the box's HBM rate
(keepdata_comb_kernel) runs at the power cap, and which clocks give way.

usage: power_probe.py            (parent: runs every case)
       power_probe.py --child CASE SECONDS
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["idle", "keepdata_table", "keepdata_gaussian", "paths_gaussian", "paths_table", "fill"]


def child(case, seconds):
    sys.path.insert(0, ROOT)
    import ctypes
    import torch
    import stock_market_monte_carlo_amd as S
    e = S.Engine(0)
    e.set_table(S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv")))
    if case == "idle":
        time.sleep(seconds)
        return
    if case == "fill":
        buf = torch.empty(1_444_000_000, dtype=torch.float32, device="cuda:0")
        fn, sync = (lambda: buf.fill_(1.0)), torch.cuda.synchronize
        bytes_per = 4.0 * buf.numel()
    elif case.startswith("keepdata"):
        mode = S.MODE_TABLE if case.endswith("table") else S.MODE_GAUSSIAN
        sim = S.Engine.make_sim(4_000_000, 360, mode, 7)
        traj, _ = e.simulate_keepdata(sim, want_final=False)
        fn = lambda: e._L.smmc_engine_simulate_keepdata(e._h, ctypes.byref(sim), ctypes.c_void_p(traj.data_ptr()), None)
        sync, bytes_per = e.sync, 4.0 * 4_000_000 * 361
    else:
        mode = S.MODE_TABLE if case.endswith("table") else S.MODE_GAUSSIAN
        sim = S.Engine.make_sim(100_000_000, 360, mode, 7)
        final = torch.empty(100_000_000, dtype=torch.float32, device="cuda:0")
        fn = lambda: e._L.smmc_engine_simulate(e._h, ctypes.byref(sim), ctypes.c_void_p(final.data_ptr()), None, None, None)
        sync, bytes_per = e.sync, 4.0e8
    fn(); sync()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        sync(); n += 20
    dt = time.time() - t0
    print(json.dumps({"case": case, "launches": n, "ms_per_launch": dt / n * 1e3, "GBps": bytes_per * n / dt / 1e9}), flush=True)


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if "power" in kl or "sclk" in kl or "mclk" in kl or "fclk" in kl or "socclk" in kl or "temperature" in kl:
                keep[k] = v
        return keep
    except Exception as ex:
        return {"error": str(ex)}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], float(sys.argv[3]))
    seconds = 8.0
    for case in CASES:
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", case, str(seconds)], stdout=subprocess.PIPE, text=True)
        time.sleep(4.0 if case != "idle" else 0.5)  # start-up and warm-up of the child
        rows = []
        while p.poll() is None:
            rows.append(sample())
            time.sleep(0.3)
        out = p.stdout.read().strip()
        print(json.dumps({"case": case, "child": json.loads(out) if out.startswith("{") else out, "samples": rows[:-1][-12:]}), flush=True)


if __name__ == "__main__":
    main()
