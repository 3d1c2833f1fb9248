"""Seeded random differential test: shapes, seeds, path offsets, parameters, tables and kernel
selections nobody wrote down by hand, each compared bit for bit with the oracle.  The cases are drawn
from a fixed numpy seed, so a failure names a reproducible case; sizes keep the oracle at a few
milliseconds per case."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases(seed, count):
    rng = np.random.default_rng(seed)
    for i in range(count):
        mode_name = ("table", "gaussian")[int(rng.integers(2))]
        t_len = int(rng.choice([1, 2, 17, 255, 1127, 2048, 2049, 5000]))
        yield dict(
            i=i, mode_name=mode_name,
            n=int(rng.choice([1, 63, 64, 255, 256, 257, 1000, 2047, 2048, 4100, 6001])) + int(rng.integers(0, 3)),
            p=int(rng.choice([0, 1, 2, 3, 4, 7, 8, 9, 31, 32, 33, 64, 100, 359, 360, 361, 500])),
            first=int(rng.choice([0, 1, 255, (1 << 32) - 300, (1 << 32), (1 << 45) + 12345, (1 << 62) - 7000])),
            seed=int(rng.integers(0, 1 << 63)) * 2 + int(rng.integers(2)),
            cap=float(rng.choice([1.0, 1000.0, 12345.678, 1e-3, 1e9])),
            mean=float(rng.choice([0.0, 0.5, -0.25, 2.0])), std=float(rng.choice([0.0, 0.83333, 4.3, 1e-3])),
            stream=int(rng.choice([2, 3, 3])), exact_div=bool(rng.integers(4) == 0),
            table=rng.normal(0.6, 4.3, t_len).astype(np.float32), n_bins=int(rng.choice([0, 1, 7, 100, 1000])),
            shift=int(rng.integers(0, 32)), kd_kernel=str(rng.choice(["", "comb", "tile"])), kd_k=int(rng.choice([1, 2, 4, 32])),
        )


def _params(oracle, S, c):
    mode = S.MODE_GAUSSIAN if c["mode_name"] == "gaussian" else S.MODE_TABLE
    hi = c["cap"] * 20.0
    sim = S.Engine.make_sim(c["n"], c["p"], mode, c["seed"], first_path=c["first"], initial_capital=c["cap"],
                            gauss_mean=c["mean"], gauss_std=c["std"], n_bins=c["n_bins"], hist_lo=0.0, hist_hi=hi,
                            exact_div=c["exact_div"], stream=c["stream"])
    op = oracle.make_params(mode, c["p"], c["n"], c["seed"], first_path=c["first"], initial_capital=c["cap"],
                            table=c["table"], gauss_mean=c["mean"], gauss_std=c["std"], n_bins=c["n_bins"], hist_lo=0.0,
                            hist_hi=hi, below_threshold=c["cap"], stream=c["stream"])
    return sim, op


def _brief(c):
    return {k: v for k, v in c.items() if k != "table"} | {"table_len": len(c["table"])}


def test_random_final_values_statistics_and_chunks(oracle):
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    try:
        for c in _cases(20261004, 120):
            eng.set_table(c["table"])
            sim, op = _params(oracle, S, c)
            r = eng.simulate(sim, want_final=True, want_chunk_stats=True, want_stats=True)
            st = eng.read_stats(r.stats_raw)
            o = oracle.counter_mc(op)
            assert np.array_equal(r.final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32)), _brief(c)
            os_ = o["stats"]
            assert (st.count, st.below, st.underflow, st.overflow) == (os_.count, os_.below, os_.underflow, os_.overflow), _brief(c)
            if c["n_bins"]:
                assert np.array_equal(st.hist, o["hist"]), _brief(c)
            assert st.min == os_.min and st.max == os_.max, _brief(c)
            assert st.sum == pytest.approx(os_.sum, rel=1e-12) and st.sumsq == pytest.approx(os_.sumsq, rel=1e-12), _brief(c)
            cm, cv = oracle.chunk_mean_var(o["final"])
            np.testing.assert_allclose(r.chunk_mean.cpu().numpy(), cm, rtol=1e-6, err_msg=str(_brief(c)))
    finally:
        eng.close()


def test_random_trajectories_every_keepdata_kernel(oracle, monkeypatch):
    """keepdata through the C ABI at a random float offset inside a guarded buffer; kernel choice left
    to the host or forced to the comb / tile form (a forced comb form silently stays with the tile
    kernel where it does not apply), rows per stream random."""
    import torch
    import stock_market_monte_carlo_amd as S
    from stock_market_monte_carlo_amd import _lib
    eng = S.Engine(0)
    try:
        for c in _cases(4102026, 90):
            if c["kd_kernel"]:
                monkeypatch.setenv("SMMC_KEEPDATA_KERNEL", c["kd_kernel"])
            else:
                monkeypatch.delenv("SMMC_KEEPDATA_KERNEL", raising=False)
            monkeypatch.setenv("SMMC_KEEPDATA_K", str(c["kd_k"]))
            eng.set_table(c["table"])
            sim, op = _params(oracle, S, c)
            n, p = c["n"], c["p"]
            buf = torch.full((n * (p + 1) + 96,), -7.0, dtype=torch.float32, device="cuda")
            lo = 32 + c["shift"]
            view = buf[lo:lo + n * (p + 1)]
            fin = torch.full((n + 8,), -7.0, dtype=torch.float32, device="cuda")
            _lib.check(eng._L.smmc_engine_simulate_keepdata(eng._h, C.byref(sim), C.c_void_p(view.data_ptr()),
                                                            C.c_void_p(fin.data_ptr())))
            eng.sync()
            got = buf.cpu().numpy()
            assert np.all(got[:lo] == -7.0) and np.all(got[lo + n * (p + 1):] == -7.0), _brief(c)
            o = oracle.counter_mc(op, want_traj=True)
            traj = got[lo:lo + n * (p + 1)].reshape(n, p + 1)
            bad = np.argwhere(traj.view(np.uint32) != o["traj"].view(np.uint32))
            assert bad.size == 0, (_brief(c), bad[:4].tolist())
            f = fin.cpu().numpy()
            assert np.all(f[n:] == -7.0) and np.array_equal(f[:n].view(np.uint32), o["final"].view(np.uint32)), _brief(c)
    finally:
        eng.close()


def test_random_host_pipeline_chunks(oracle, monkeypatch):
    """simulate_to_host with a small random chunk length: several chunks, ragged tail, statistics merged
    over chunks, pageable and pinned destinations."""
    import torch
    import stock_market_monte_carlo_amd as S
    rng = np.random.default_rng(99)
    for c in _cases(777, 24):
        chunk = int(rng.choice([256, 512, 1024, 4096]))
        monkeypatch.setenv("SMMC_HOST_CHUNK_PATHS", str(chunk))
        eng = S.Engine(0)  # the chunk length is read when an engine is created
        try:
            eng.set_table(c["table"])
            sim, op = _params(oracle, S, c)
            out = torch.empty(c["n"], dtype=torch.float32, pin_memory=bool(rng.integers(2))).numpy()
            host, st, _ = eng.simulate_to_host(sim, out=out, want_stats=True)
            o = oracle.counter_mc(op)
            assert np.array_equal(host.view(np.uint32), o["final"].view(np.uint32)), (_brief(c), chunk)
            assert st.count == c["n"] and st.below == o["stats"].below, (_brief(c), chunk)
            if c["n_bins"]:
                assert np.array_equal(st.hist, o["hist"]), (_brief(c), chunk)
        finally:
            eng.close()


def test_random_reference_stream_cases(oracle):
    """The reference CPU engine's own stream (SMMC_FLAG_STREAM_REF) on 80 random cases: path counts, lengths on both
    sides of the windowed kernel's limits, table sizes from 1 entry (every draw index 0) to 16384 (no rejections at
    all: 2^32 is a multiple of it), seeds and path offsets that wrap past 2^32, capitals, both divides -- final values
    and, every fourth case, whole trajectories, bit for bit against oracle engine (R) and its index stream."""
    import stock_market_monte_carlo_amd as S
    rng = np.random.default_rng(3035)
    eng = S.Engine(0)
    try:
        for i in range(80):
            t_len = int(rng.choice([1, 2, 3, 17, 255, 1127, 2048, 4097, 12289, 16384]))
            table = rng.normal(0.6, 4.3, t_len).astype(np.float32)
            n = int(rng.choice([1, 63, 64, 65, 255, 256, 257, 1000, 2049, 5003]))
            p = int(rng.choice([0, 1, 2, 3, 31, 32, 33, 226, 227, 228, 360, 453, 454, 455, 700]))
            seed = int(rng.integers(0, 1 << 63))
            first = int(rng.choice([0, 1, 255, (1 << 32) - 300, (1 << 32), (1 << 45) + 12345]))
            if i % 5 == 0:  # the per-path seed (seed + first + id) mod 2^32 wraps in the middle of the launch
                seed = ((1 << 32) - n // 2 - first) % (1 << 64)
            cap = float(rng.choice([1.0, 1000.0, 12345.678, 1e9]))
            exact = bool(rng.integers(3) == 0)
            eng.set_table(table)
            sim = S.Engine.make_sim(n, p, S.MODE_TABLE, seed, first_path=first, initial_capital=cap, exact_div=exact,
                                    stream="ref")
            brief = dict(i=i, t_len=t_len, n=n, p=p, seed=seed, first=first, cap=cap, exact=exact)
            seed0 = (seed + first) & 0xFFFFFFFF
            want, _ = oracle.ref_mc_simulations(n, p, cap, table, seed0)
            if i % 4 == 3:
                traj, fin = eng.simulate_keepdata(sim)
                eng.sync()
                got, rows = fin.cpu().numpy(), traj.cpu().numpy()
                for k in {0, n // 2, n - 1}:
                    idx = oracle.mt19937_indices((seed0 + k) & 0xFFFFFFFF, t_len, p)
                    assert np.array_equal(rows[k].view(np.uint32), oracle.many_updates(cap, table[idx], p).view(np.uint32)), (brief, k)
            else:
                r = eng.simulate(sim)
                eng.sync()
                got = r.final.cpu().numpy()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), brief
    finally:
        eng.close()
