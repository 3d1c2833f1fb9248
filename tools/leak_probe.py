"""Resident set and free device memory every 20 cycles of engine / group create-simulate-destroy (200 cycles each):
the runtime's pools step up within the first cycles of a new pattern, then stay flat (profiles/r03/leak_probe.txt)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stock_market_monte_carlo_amd as S
from bench import load_table
from tools.soak import rss_bytes
table = load_table()
sim = S.Engine.make_sim(300_001, 12, S.MODE_TABLE, 3, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
host = np.empty(300_001, dtype=np.float32)
def trace(name, fn, cycles=200, every=20):
    pts = []
    for i in range(cycles + 1):
        if i % every == 0:
            pts.append((i, rss_bytes() >> 20, torch.cuda.mem_get_info()[0] >> 20))
        fn()
    print(name, pts, flush=True)
def grp(devs):
    def run():
        g = S.Group(devs, merge="host"); g.set_table(table); g.simulate(sim, out=host, want_final=True, want_stats=True); g.close()
    return run
def eng_cycle():
    e = S.Engine(0, stream="new"); e.set_table(table); e.simulate_to_host(sim, out=host, want_stats=True); e.close()
def grp_nosim():
    g = S.Group([0, 0, 0], merge="host"); g.close()
trace("engine+to_host", eng_cycle)
trace("group[0]", grp([0]))
trace("group[0,0,0] no simulate", grp_nosim)
trace("group[0,0,0]", grp([0, 0, 0]))
trace("group[0,0,0] again", grp([0, 0, 0]))
