import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import stock_market_monte_carlo_amd as S
from oracle import oracle as O
table = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
eng = S.Engine(0); eng.set_table(table)
for mode, omode in ((S.MODE_TABLE, O.MODE_TABLE), (S.MODE_GAUSSIAN, O.MODE_GAUSSIAN)):
    sim = S.Engine.make_sim(4099, 360, mode, 2024, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
    r = eng.simulate(sim, want_final=True, want_chunk_stats=True, want_stats=True)
    st = eng.read_stats(r.stats_raw)
    o = O.counter_mc(O.make_params(omode, 360, 4099, 2024, table=table, n_bins=100, hist_lo=0.0, hist_hi=20000.0))
    os_ = o["stats"]
    print("final equal", np.array_equal(r.final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32)))
    print("gpu ", st.count, st.below, st.underflow, st.overflow, st.sum, st.sumsq, st.min, st.max)
    print("orc ", os_.count, os_.below, os_.underflow, os_.overflow, os_.sum, os_.sumsq, os_.min, os_.max)
    print("hist gpu", st.hist[:12], st.hist.sum())
    print("hist orc", o["hist"][:12], o["hist"].sum())
    raw = r.stats_raw.cpu().numpy()
    print(raw[:64].view(np.uint64))
