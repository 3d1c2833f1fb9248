"""Soak of every entry a long-lived caller loops over: device memory and host RSS before / after many calls.

    python tools/soak.py [--scale 1.0] > gpurun_out/soak.jsonl

One JSON line per case: calls, seconds, free device memory (hipMemGetInfo through torch) and the process's
resident set at the loop's mid-point and at its end (the first half lets lazily created buffers and the runtime's pools settle).
A leak of one staging buffer or one registration per call shows as a slope; the asserts at the end fail the
run when any case lost more than 8 MiB of device memory or grew more than 64 MiB of RSS.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def rss_bytes():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="multiplies every case's call count")
    args = ap.parse_args()
    import torch
    import stock_market_monte_carlo_amd as S
    from bench import load_table

    table = load_table()
    eng = S.Engine(0)
    eng.set_table(table)
    results = []

    def case(name, calls, fn, warm=2):
        calls = max(1, int(calls * args.scale))
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        # The runtime's own pools (signals, kernel arguments, per-thread state) grow to their high-water mark in
        # the first cycles of a new pattern and stay there (tools/leak_probe.py: flat from cycle 20 to 200),
        # so the slope is taken over the SECOND half of the loop.
        t0 = time.perf_counter()
        for _ in range(calls // 2):
            fn()
        torch.cuda.synchronize()
        free0, rss0 = torch.cuda.mem_get_info()[0], rss_bytes()
        for _ in range(calls - calls // 2):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        free1, rss1 = torch.cuda.mem_get_info()[0], rss_bytes()
        line = {"case": name, "calls": calls, "seconds": round(dt, 3), "device_free_mid": free0, "device_free_after": free1,
                "device_lost_bytes": free0 - free1, "rss_mid": rss0, "rss_after": rss1, "rss_grown_bytes": rss1 - rss0}
        results.append(line)
        print(json.dumps(line), flush=True)

    g = S.Engine.make_sim(10_000_000, 360, S.MODE_GAUSSIAN, 11, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
    t = S.Engine.make_sim(10_000_000, 360, S.MODE_TABLE, 11, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
    r = S.Engine.make_sim(4_000_000, 360, S.MODE_TABLE, 11, n_bins=100, hist_lo=0.0, hist_hi=20000.0, stream="ref")
    rl = S.Engine.make_sim(400_000, 1000, S.MODE_TABLE, 11, n_bins=100, hist_lo=0.0, hist_hi=20000.0, stream="ref")
    rg = S.Engine.make_sim(400_000, 2000, S.MODE_TABLE, 11, n_bins=100, hist_lo=0.0, hist_hi=20000.0, stream="ref")
    k = S.Engine.make_sim(300_000, 360, S.MODE_GAUSSIAN, 11)
    out = torch.empty(10_000_000, dtype=torch.float32, device=eng.tdevice)
    host = torch.empty(10_000_000, dtype=torch.float32, pin_memory=True).numpy()
    pageable = np.empty(10_000_000, dtype=np.float32)

    case("simulate gaussian, all outputs, caller's tensor", 1500,
         lambda: eng.simulate(g, want_final=True, want_chunk_stats=True, want_stats=True, out=out))
    case("simulate table, engine-allocated outputs", 1500,
         lambda: eng.simulate(t, want_final=True, want_chunk_stats=True, want_stats=True))
    case("simulate reference stream (windowed + redo)", 400, lambda: eng.simulate(r, want_final=True, want_stats=True))
    case("simulate reference stream (tree kernel, 1000 periods)", 200, lambda: eng.simulate(rl, want_final=True, want_stats=True))
    case("simulate reference stream (generic kernel, 2000 periods)", 200, lambda: eng.simulate(rg, want_final=True, want_stats=True))
    case("simulate_to_host, pinned result", 400, lambda: eng.simulate_to_host(g, out=host, want_stats=True))
    for policy in ("whole", "chunk", "0"):  # read when an engine is created
        os.environ["SMMC_PIN_HOST"] = policy
        pe = S.Engine(0)
        pe.set_table(table)
        case(f"simulate_to_host, pageable result, SMMC_PIN_HOST={policy}", 200, lambda: pe.simulate_to_host(g, out=pageable))
        pe.close()
    del os.environ["SMMC_PIN_HOST"]
    case("simulate_keepdata (433 MB of trajectories per call)", 300, lambda: eng.simulate_keepdata(k))
    vals = eng.simulate(g, want_final=True).final
    case("values_stats + quartiles", 1000, lambda: (eng.values_stats(vals, n_bins=100, hist_lo=0.0, hist_hi=20000.0), eng.quartiles(vals)))

    def engine_cycle():
        e = S.Engine(0, stream="new")
        e.set_table(table)
        e.simulate(S.Engine.make_sim(100_000, 12, S.MODE_TABLE, 3), want_stats=True)
        e.close()

    case("engine create / simulate / destroy", 200, engine_cycle)

    def group_cycle(merge):
        def run():
            grp = S.Group([0] if merge == "rccl" else [0, 0, 0], merge=merge)
            grp.set_table(table)
            grp.simulate(S.Engine.make_sim(300_001, 12, S.MODE_TABLE, 3, n_bins=100, hist_lo=0.0, hist_hi=20000.0),
                         want_final=True, want_stats=True)
            grp.close()
        return run

    case("group of three shards: create / simulate / destroy (host merge)", 100, group_cycle("host"))
    case("group of one: create / simulate / destroy (RCCL communicator per cycle)", 16, group_cycle("rccl"), warm=1)
    grp = S.Group([0, 0, 0], merge="host")
    grp.set_table(table)
    case("group of three shards, resident: simulate to the caller's array", 300,
         lambda: grp.simulate(g, out=pageable, want_final=True, want_stats=True))
    grp.close()

    bad = [x for x in results if x["device_lost_bytes"] > (8 << 20) or x["rss_grown_bytes"] > (64 << 20)]
    print(json.dumps({"cases": len(results), "over_limit": [x["case"] for x in bad]}), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
