"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bit-exact for final values (binary32 patterns), histogram bucket counts and the
below/underflow/overflow counters; sums (double) to 1e-12 relative because the
device adds in a tree order; per-chunk mean/variance (float) to 1e-6 relative.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 0x5EED0123456789AB


@pytest.fixture(scope="module")
def eng(table):
    import stock_market_monte_carlo_amd as S
    e = S.Engine(0)
    e.set_table(table)
    yield e
    e.close()


def _modes():
    from stock_market_monte_carlo_amd import MODE_GAUSSIAN, MODE_TABLE
    return {"table": MODE_TABLE, "gaussian": MODE_GAUSSIAN}


def _run_both(eng, oracle, table, mode, n, p, first=0, seed=SEED, n_bins=0, lo=0.0, hi=1.0, cap=1000.0,
              mean=0.5, std=0.83333, exact_div=False, below=None):
    from stock_market_monte_carlo_amd import Engine
    sim = Engine.make_sim(n, p, mode, seed, first_path=first, initial_capital=cap, gauss_mean=mean, gauss_std=std,
                          n_bins=n_bins, hist_lo=lo, hist_hi=hi, exact_div=exact_div, below_threshold=below)
    r = eng.simulate(sim, want_final=True, want_chunk_stats=True, want_stats=True)
    st = eng.read_stats(r.stats_raw)
    op = oracle.make_params(mode, p, n, seed, first_path=first, initial_capital=cap, table=table, gauss_mean=mean,
                            gauss_std=std, n_bins=n_bins, hist_lo=lo, hist_hi=hi,
                            below_threshold=cap if below is None else below)
    o = oracle.counter_mc(op)
    return r, st, o


@pytest.mark.parametrize("mode_name", ["table", "gaussian"])
@pytest.mark.parametrize("p", [0, 1, 3, 4, 5, 7, 8, 9, 15, 360, 1000])
def test_final_values_bit_exact(eng, oracle, table, mode_name, p):
    mode = _modes()[mode_name]
    n = 5000 + 37  # ragged: not a multiple of 256
    r, st, o = _run_both(eng, oracle, table, mode, n, p, n_bins=100, lo=0.0, hi=20000.0)
    got = r.final.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), o["final"].view(np.uint32))
    os_ = o["stats"]
    assert st.count == n == os_.count
    assert (st.below, st.underflow, st.overflow) == (os_.below, os_.underflow, os_.overflow)
    assert np.array_equal(st.hist, o["hist"])
    assert int(st.hist.sum()) + st.underflow + st.overflow == n
    assert st.min == os_.min and st.max == os_.max
    assert st.sum == pytest.approx(os_.sum, rel=1e-12)
    assert st.sumsq == pytest.approx(os_.sumsq, rel=1e-12)
    cm, cv = oracle.chunk_mean_var(o["final"])
    np.testing.assert_allclose(r.chunk_mean.cpu().numpy(), cm, rtol=1e-6)
    np.testing.assert_allclose(r.chunk_var.cpu().numpy(), cv, rtol=1e-5, atol=1e-6 * float(np.max(cv) + 1))


@pytest.mark.parametrize("mode_name", ["table", "gaussian"])
def test_path_ids_beyond_32_bits_and_seed_halves(eng, oracle, table, mode_name):
    mode = _modes()[mode_name]
    for first, seed in [((1 << 32) - 100, SEED), ((1 << 40) + 12345, 1), (7, 0xFFFFFFFF00000000)]:
        r, _, o = _run_both(eng, oracle, table, mode, 700, 36, first=first, seed=seed)
        assert np.array_equal(r.final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32)), (first, seed)


@pytest.mark.parametrize("stream", [2, 3])
def test_golden_counter_stream(eng, table, stream):
    """Frozen oracle outputs (tests/golden/counter_stream_v{2,3}.json, made by make_golden.py): v3 is
    the default Gaussian draw, v2 (round 1's) is selected with SMMC_FLAG_STREAM_V2."""
    from stock_market_monte_carlo_amd import Engine
    with open(os.path.join(HERE, "golden", f"counter_stream_v{stream}.json")) as f:
        gold = json.load(f)
    for c in gold["cases"]:
        sim = Engine.make_sim(c["n_paths"], c["n_periods"], _modes()[c["mode"]], c["seed"], first_path=c["first_path"],
                              initial_capital=c["initial_capital"], gauss_mean=c["gauss_mean"], gauss_std=c["gauss_std"],
                              n_bins=c["n_bins"], hist_lo=c["hist_lo"], hist_hi=c["hist_hi"],
                              below_threshold=c["below_threshold"], stream=stream)
        r = eng.simulate(sim, want_stats=True)
        st = eng.read_stats(r.stats_raw)
        assert [int(x) for x in r.final.cpu().numpy().view(np.uint32)] == c["final_bits"], (c["mode"], c["n_periods"])
        assert [int(x) for x in st.hist] == c["hist"]
        assert (st.below, st.underflow, st.overflow) == (c["below"], c["underflow"], c["overflow"])


def test_stream_v2_stays_selectable_in_every_kernel(eng, oracle, table, monkeypatch):
    """SMMC_FLAG_STREAM_V2: round 1's Gaussian draw through the paths kernel (fast, checked and IEEE
    divide), both keepdata kernels and the host pipeline, bit-exact against the oracle's v2 engine."""
    from stock_market_monte_carlo_amd import Engine, MODE_GAUSSIAN
    for n, p, kw in ((3000, 360, {}), (1000, 1000, {}), (777, 37, {"exact_div": True}),
                     (2000, 360, {"gauss_mean": 2.0, "gauss_std": 9.0})):  # the last one: range-checked divide
        sim = Engine.make_sim(n, p, MODE_GAUSSIAN, SEED, first_path=5, n_bins=64, hist_lo=0.0, hist_hi=30000.0, stream=2, **kw)
        r = eng.simulate(sim, want_stats=True)
        st = eng.read_stats(r.stats_raw)
        op = oracle.make_params(oracle.MODE_GAUSSIAN, p, n, SEED, first_path=5, n_bins=64, hist_lo=0.0, hist_hi=30000.0,
                                stream=2, gauss_mean=kw.get("gauss_mean", 0.5), gauss_std=kw.get("gauss_std", 0.83333))
        o = oracle.counter_mc(op)
        assert np.array_equal(r.final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32)), (n, p, kw)
        assert np.array_equal(st.hist, o["hist"]) and st.below == o["stats"].below
        # and v3 on the same request really is another stream
        sim3 = Engine.make_sim(n, p, MODE_GAUSSIAN, SEED, first_path=5, **kw)
        assert not np.array_equal(eng.simulate(sim3).final.cpu().numpy(), o["final"])
    for kernel, n, p in (("tile", 700, 360), ("comb", 2048 + 300, 64)):
        monkeypatch.setenv("SMMC_KEEPDATA_KERNEL", kernel)
        sim = Engine.make_sim(n, p, MODE_GAUSSIAN, SEED, first_path=11, stream=2)
        traj, final = eng.simulate_keepdata(sim)
        o = oracle.counter_mc(oracle.make_params(oracle.MODE_GAUSSIAN, p, n, SEED, first_path=11, stream=2), want_traj=True)
        assert np.array_equal(traj.cpu().numpy().view(np.uint32), o["traj"].view(np.uint32)), kernel
        assert np.array_equal(final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32))
    monkeypatch.delenv("SMMC_KEEPDATA_KERNEL")
    sim = Engine.make_sim(5000, 24, MODE_GAUSSIAN, SEED, stream=2)
    host, _, _ = eng.simulate_to_host(sim)
    o = oracle.counter_mc(oracle.make_params(oracle.MODE_GAUSSIAN, 24, 5000, SEED, stream=2))
    assert np.array_equal(host.view(np.uint32), o["final"].view(np.uint32))


@pytest.mark.parametrize("mode_name", ["table", "gaussian"])
def test_exact_divide_variant_matches_fast_variant(eng, oracle, table, mode_name):
    mode = _modes()[mode_name]
    a, _, o = _run_both(eng, oracle, table, mode, 3000, 360, exact_div=False)
    b, _, _ = _run_both(eng, oracle, table, mode, 3000, 360, exact_div=True)
    assert np.array_equal(a.final.cpu().numpy().view(np.uint32), b.final.cpu().numpy().view(np.uint32))
    assert np.array_equal(b.final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32))


def test_out_of_range_trajectories_take_the_ieee_divide(eng, oracle):
    """Huge swings push values past 2^100 / to inf and 0: the host must pick the exact-divide
    kernel, and results must still equal the oracle bit for bit (inf and 0 included)."""
    from stock_market_monte_carlo_amd import MODE_TABLE
    wild = np.array([9000.0, -99.9, 50000.0, -100.0, 3.0e6, -50.0, 10.0], dtype=np.float32)
    eng.set_table(wild)
    try:
        r, st, o = _run_both(eng, oracle, wild, MODE_TABLE, 2000, 60, cap=1.0e-20, n_bins=10, lo=0.0, hi=1.0e30)
        got = r.final.cpu().numpy()
        assert np.array_equal(got.view(np.uint32), o["final"].view(np.uint32))
        assert np.array_equal(st.hist, o["hist"]) and st.overflow == o["stats"].overflow
    finally:
        from conftest import load_table
        eng.set_table(load_table())


def test_fast_divide_window_edges(eng, oracle):
    """The host picks the reciprocal-multiply divide while every product total * a stays inside
    [2^-89, 2^127) and the IEEE divide beyond: both sides of both edges equal the oracle (which
    always divides), down to subnormal and up to inf results."""
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE
    from conftest import load_table
    try:
        for ret, periods in ((60.0, (160, 170, 180, 200)), (-60.0, (60, 66, 67, 70, 95, 110))):
            tab = np.array([ret], dtype=np.float32)
            eng.set_table(tab)
            for p in periods:
                r, st, o = _run_both(eng, oracle, tab, MODE_TABLE, 300, p, cap=1.0, n_bins=8, lo=0.0, hi=1.0e38)
                assert np.array_equal(r.final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32)), (ret, p)
                assert np.array_equal(st.hist, o["hist"])
                traj, _ = eng.simulate_keepdata(Engine.make_sim(70, p, MODE_TABLE, SEED, initial_capital=1.0))
                ot = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, p, 70, SEED, table=tab, initial_capital=1.0), want_traj=True)
                assert np.array_equal(traj.cpu().numpy().view(np.uint32), ot["traj"].view(np.uint32)), (ret, p)
    finally:
        eng.set_table(load_table())


def test_checked_divide_for_tables_that_cannot_be_proven_safe(eng, oracle, table):
    """A table with a +42 % month cannot be PROVEN to stay in the fast divide's domain over 360
    periods, yet no path gets there: the engine then runs the fast divide with one range check per
    Philox block and redoes a path that leaves the window with the IEEE divide.  Same bits as the
    oracle (which always divides) without offenders, with every second path offending upwards
    (to inf) and downwards (to subnormals and 0)."""
    from stock_market_monte_carlo_amd import Engine, MODE_GAUSSIAN, MODE_TABLE, _lib
    from conftest import load_table
    assert eng.divide_kind(Engine.make_sim(1000, 360, MODE_TABLE, SEED)) == _lib.DIV_FAST
    assert eng.divide_kind(Engine.make_sim(1000, 1000, MODE_GAUSSIAN, SEED)) == _lib.DIV_FAST
    assert eng.divide_kind(Engine.make_sim(1000, 1000, MODE_TABLE, SEED)) == _lib.DIV_CHECKED
    assert eng.divide_kind(Engine.make_sim(1000, 1000, MODE_TABLE, SEED), keepdata=True) == _lib.DIV_EXACT
    assert eng.divide_kind(Engine.make_sim(1000, 360, MODE_TABLE, SEED, exact_div=True)) == _lib.DIV_EXACT
    try:
        real = table.copy()
        real[7], real[100] = 42.2, -29.7  # the S&P 500's best and worst months
        eng.set_table(real)
        sim = Engine.make_sim(20_000, 360, MODE_TABLE, SEED)
        assert eng.divide_kind(sim) == _lib.DIV_CHECKED
        for p in (360, 1000, 7, 8, 9):
            r, st, o = _run_both(eng, oracle, real, MODE_TABLE, 20_000, p, n_bins=100, lo=0.0, hi=1.0e6)
            assert np.array_equal(r.final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32)), p
            assert np.array_equal(st.hist, o["hist"])
        for tab, cap in (([60.0, -20.0], 2.0 ** 50), ([60.0, -50.0], 2.0 ** -60)):
            tab = np.array(tab, dtype=np.float32)
            eng.set_table(tab)
            for p in (400, 403):
                assert eng.divide_kind(Engine.make_sim(10, p, MODE_TABLE, SEED, initial_capital=cap)) == _lib.DIV_CHECKED
                r, st, o = _run_both(eng, oracle, tab, MODE_TABLE, 6000, p, cap=cap, n_bins=16, lo=0.0, hi=1.0e38)
                got, want = r.final.cpu().numpy(), o["final"]
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (tab, p)
                assert np.array_equal(st.hist, o["hist"])
                out = np.isinf(want) | (want < 2.0 ** -110) if cap > 1 else (want < 2.0 ** -118)
                assert 0.1 < out.mean() < 0.9  # the rerun really is exercised, and so is its absence
    finally:
        eng.set_table(load_table())


def test_histogram_edges_and_single_bin(eng, oracle, table):
    from stock_market_monte_carlo_amd import MODE_TABLE
    # narrow range: most values under/overflow; 1 bin; 4096 bins
    for n_bins, lo, hi in [(1, 900.0, 1100.0), (4096, 0.0, 50000.0), (7, 5000.0, 5000.5), (100, -10.0, 1.0)]:
        _, st, o = _run_both(eng, oracle, table, MODE_TABLE, 4000, 120, n_bins=n_bins, lo=lo, hi=hi, below=5000.0)
        assert np.array_equal(st.hist, o["hist"]), (n_bins, lo, hi)
        assert (st.below, st.underflow, st.overflow) == (o["stats"].below, o["stats"].underflow, o["stats"].overflow)


def test_sharding_is_invisible(eng, table):
    """A path's value depends on (seed, global id) only: shards concatenate to the whole."""
    from stock_market_monte_carlo_amd import Engine, MODE_GAUSSIAN
    from stock_market_monte_carlo_amd.engine import merge_stats_bytes, stats_from_bytes
    n = 100003
    whole = eng.simulate(Engine.make_sim(n, 48, MODE_GAUSSIAN, 99, n_bins=64, hist_lo=0, hist_hi=5000), want_stats=True)
    parts, recs = [], []
    cuts = [0, 1, 256, 33333, 33334, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        r = eng.simulate(Engine.make_sim(b - a, 48, MODE_GAUSSIAN, 99, first_path=a, n_bins=64, hist_lo=0, hist_hi=5000),
                         want_stats=True)
        parts.append(r.final.cpu().numpy())
        recs.append(r.stats_raw.cpu().numpy().tobytes())
    assert np.array_equal(np.concatenate(parts).view(np.uint32), whole.final.cpu().numpy().view(np.uint32))
    merged = stats_from_bytes(merge_stats_bytes(recs))
    w = eng.read_stats(whole.stats_raw)
    assert merged.count == w.count == n and np.array_equal(merged.hist, w.hist)
    assert (merged.below, merged.underflow, merged.overflow) == (w.below, w.underflow, w.overflow)
    assert merged.min == w.min and merged.max == w.max
    assert merged.sum == pytest.approx(w.sum, rel=1e-12)


def test_empty_and_tiny(eng, table):
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE
    r = eng.simulate(Engine.make_sim(0, 360, MODE_TABLE, 1, n_bins=10, hist_lo=0, hist_hi=10), want_stats=True,
                     want_chunk_stats=True)
    st = eng.read_stats(r.stats_raw)
    assert st.count == 0 and st.hist.sum() == 0 and st.min == float("inf") and st.max == float("-inf")
    assert r.final.numel() == 0 and r.chunk_mean.numel() == 0
    r = eng.simulate(Engine.make_sim(1, 0, MODE_TABLE, 1, initial_capital=123.5))
    assert r.final.cpu().numpy().tolist() == [123.5]


def test_large_table_uses_the_sparse_schedule(eng, oracle):
    """> 2048 entries: four draws per Philox block (the dense digit extraction would bias)."""
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE
    from conftest import load_table
    rng = np.random.default_rng(11)
    big = rng.normal(0.05, 1.1, 5000).astype(np.float32)
    eng.set_table(big)
    try:
        for p in (1, 7, 8, 250):
            r, st, o = _run_both(eng, oracle, big, MODE_TABLE, 3001, p, n_bins=20, lo=0.0, hi=3000.0)
            assert np.array_equal(r.final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32)), p
            assert np.array_equal(st.hist, o["hist"])
        for n, p in ((200, 70), (1500, 200), (700, 63), (333, 360), (129, 31)):
            sim = Engine.make_sim(n, p, MODE_TABLE, SEED, first_path=5)
            traj, _ = eng.simulate_keepdata(sim)
            o = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, p, n, SEED, first_path=5, table=big), want_traj=True)
            assert np.array_equal(traj.cpu().numpy().view(np.uint32), o["traj"].view(np.uint32)), (n, p)
    finally:
        eng.set_table(load_table())


def test_maximum_table_and_histogram(eng, oracle):
    """SMMC_MAX_TABLE entries (64 KiB of LDS) with SMMC_MAX_BINS buckets: above the default
    64 KiB dynamic-LDS limit, so the launch has to opt in."""
    from stock_market_monte_carlo_amd import MODE_TABLE
    from conftest import load_table
    rng = np.random.default_rng(12)
    big = rng.normal(0.05, 1.0, 16384).astype(np.float32)
    eng.set_table(big)
    try:
        r, st, o = _run_both(eng, oracle, big, MODE_TABLE, 2000, 50, n_bins=4096, lo=0.0, hi=3000.0)
        assert np.array_equal(r.final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32))
        assert np.array_equal(st.hist, o["hist"])
        # keepdata next to a 64 KiB table: fewer waves fit a workgroup
        from stock_market_monte_carlo_amd import Engine
        traj, _ = eng.simulate_keepdata(Engine.make_sim(900, 100, MODE_TABLE, SEED))
        o = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 100, 900, SEED, table=big), want_traj=True)
        assert np.array_equal(traj.cpu().numpy().view(np.uint32), o["traj"].view(np.uint32))
    finally:
        eng.set_table(load_table())


def test_table_of_one_entry_is_deterministic_compounding(eng, oracle):
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE
    from conftest import load_table
    eng.set_table(np.array([1.5], dtype=np.float32))
    try:
        r = eng.simulate(Engine.make_sim(300, 25, MODE_TABLE, 5))
        want = oracle.many_updates(1000.0, np.full(25, 1.5, dtype=np.float32), 25)[-1]
        assert np.all(r.final.cpu().numpy().view(np.uint32) == np.float32(want).view(np.uint32))
    finally:
        eng.set_table(load_table())


@pytest.mark.parametrize("mode_name", ["table", "gaussian"])
def test_keepdata_trajectories_bit_exact(eng, oracle, table, mode_name):
    from stock_market_monte_carlo_amd import Engine
    mode = _modes()[mode_name]
    for n, p in [(300, 360), (1000, 1), (257, 64), (64, 65), (700, 130), (130, 31), (65, 7), (1, 100),
                 (500, 63), (500, 62), (129, 95), (3, 200), (64, 127), (4097, 70), (2, 63), (1000, 1000),
                 (3, (1 << 22) + 5)]:  # the last one: rows too long for 32-bit byte offsets
        sim = Engine.make_sim(n, p, mode, SEED, first_path=11)
        traj, final = eng.simulate_keepdata(sim)
        op = oracle.make_params(mode, p, n, SEED, first_path=11, table=table)
        o = oracle.counter_mc(op, want_traj=True)
        assert np.array_equal(traj.cpu().numpy().view(np.uint32), o["traj"].view(np.uint32)), (n, p)
        assert np.array_equal(final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32))
        # the trajectory IS many_updates of the path's draws (src/simulations.cpp:175-186)
        rets = oracle.counter_path_returns(op, 11 + n - 1)
        assert np.array_equal(oracle.many_updates(1000.0, rets, p).view(np.uint32),
                              traj[n - 1].cpu().numpy().view(np.uint32))


def test_keepdata_at_any_base_alignment_and_zero_periods(eng, oracle, table):
    """The C ABI takes any 4-byte aligned d_traj: the row phases follow the base address.  And
    n_periods = 0 is one column of initial capital."""
    import ctypes as C
    import torch
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE, _lib
    for shift in (1, 2, 3, 5, 31):
        for n, p in ((200, 360), (129, 64), (70, 33)):
            sim = Engine.make_sim(n, p, MODE_TABLE, SEED, first_path=3)
            buf = torch.full((n * (p + 1) + 64,), -1.0, dtype=torch.float32, device="cuda")
            view = buf[shift:shift + n * (p + 1)]
            _lib.check(eng._L.smmc_engine_simulate_keepdata(eng._h, C.byref(sim), C.c_void_p(view.data_ptr()), None))
            eng.sync()
            o = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, p, n, SEED, first_path=3, table=table), want_traj=True)
            got = buf.cpu().numpy()
            assert np.array_equal(got[shift:shift + n * (p + 1)].view(np.uint32), o["traj"].reshape(-1).view(np.uint32)), (shift, n, p)
            assert np.all(got[:shift] == -1.0) and np.all(got[shift + n * (p + 1):] == -1.0)  # nothing outside
    traj, final = eng.simulate_keepdata(Engine.make_sim(300, 0, MODE_TABLE, SEED))
    assert traj.shape == (300, 1) and bool((traj == 1000.0).all()) and bool((final == 1000.0).all())
    # rejected before any launch: a pointer that is not 4-byte aligned, too many periods
    sim = Engine.make_sim(10, 5, MODE_TABLE, SEED)
    assert eng._L.smmc_engine_simulate_keepdata(eng._h, C.byref(sim), C.c_void_p(traj.data_ptr() + 2), None) == -1  # SMMC_ERR_INVALID
    sim = Engine.make_sim(10, 1 << 24, MODE_TABLE, SEED)
    assert eng._L.smmc_engine_simulate_keepdata(eng._h, C.byref(sim), C.c_void_p(traj.data_ptr()), None) == -1


@pytest.mark.parametrize("knobs", [{"SMMC_KEEPDATA_TILE": "16"}, {"SMMC_KEEPDATA_WAVES": "1"},
                                   {"SMMC_KEEPDATA_WAVES": "7"}, {"SMMC_KEEPDATA_WAVES": "12"}, {"SMMC_KEEPDATA_WAVES": "4"},
                                   {"SMMC_KEEPDATA_TILE": "16", "SMMC_KEEPDATA_WAVES": "9"}])
def test_keepdata_tuning_knobs_do_not_change_results(eng, oracle, table, knobs, monkeypatch):
    """Tile width and workgroup size are read from the environment per call; every setting writes
    the same bits."""
    from stock_market_monte_carlo_amd import Engine
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for mode_name, mode in _modes().items():
        for n, p in ((700, 360), (257, 64), (130, 31), (4097, 70), (333, 129)):
            sim = Engine.make_sim(n, p, mode, SEED, first_path=11)
            traj, final = eng.simulate_keepdata(sim)
            o = oracle.counter_mc(oracle.make_params(mode, p, n, SEED, first_path=11, table=table), want_traj=True)
            assert np.array_equal(traj.cpu().numpy().view(np.uint32), o["traj"].view(np.uint32)), (knobs, mode_name, n, p)
            assert np.array_equal(final.cpu().numpy().view(np.uint32), o["final"].view(np.uint32))


def test_simulate_to_host_pipeline_matches_device_path(eng, table):
    """More than one 16 Mi-path chunk through the overlapped D2H pipeline."""
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE
    n = (1 << 24) * 2 + 12345
    sim = Engine.make_sim(n, 4, MODE_TABLE, 3, n_bins=50, hist_lo=800, hist_hi=1300)
    host, st, (cm, cv) = eng.simulate_to_host(sim, want_stats=True, want_chunk_stats=True)
    dev = eng.simulate(sim, want_stats=True, want_chunk_stats=True)
    assert np.array_equal(host.view(np.uint32), dev.final.cpu().numpy().view(np.uint32))
    assert np.array_equal(cm, dev.chunk_mean.cpu().numpy()) and np.array_equal(cv, dev.chunk_var.cpu().numpy())
    w = eng.read_stats(dev.stats_raw)
    assert st.count == n and np.array_equal(st.hist, w.hist) and st.below == w.below
    assert st.sum == pytest.approx(w.sum, rel=1e-12)


def test_full_size_properties(eng, table):
    """BASELINE configs 2/3 size (1e8 paths x 360): properties that need no oracle."""
    from stock_market_monte_carlo_amd import Engine, MODE_GAUSSIAN, MODE_TABLE
    n = 100_000_000
    for mode in (MODE_TABLE, MODE_GAUSSIAN):
        sim = Engine.make_sim(n, 360, mode, SEED, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
        a = eng.simulate(sim, want_final=True, want_stats=True)
        st = eng.read_stats(a.stats_raw)
        assert st.count == n and int(st.hist.sum()) + st.underflow + st.overflow == n
        # the statistics kernel path and the final values agree
        fin = a.final
        assert int((fin < 1000.0).sum().item()) == st.below
        assert float(fin.min().item()) == st.min and float(fin.max().item()) == st.max
        assert float(fin.double().sum().item()) == pytest.approx(st.sum, rel=1e-11)
        # run-to-run determinism (integer checksum of all bit patterns)
        b = eng.simulate(sim, want_final=True)
        import torch
        assert torch.equal(a.final.view(torch.int32), b.final.view(torch.int32))
        # spot check against the oracle: the last 300 paths
        del b
    from oracle import oracle as O
    sim = Engine.make_sim(n, 360, MODE_TABLE, SEED)
    a = eng.simulate(sim)
    tail = a.final[-300:].cpu().numpy()
    op = O.make_params(O.MODE_TABLE, 360, 300, SEED, first_path=n - 300, table=table)
    assert np.array_equal(tail.view(np.uint32), O.counter_mc(op)["final"].view(np.uint32))


def test_more_than_2_to_32_paths_in_one_launch(eng, oracle, table):
    """Statistics only (no 17 GB result array): 64-bit path counts and ids inside one launch."""
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE
    n = (1 << 32) + 12345
    sim = Engine.make_sim(n, 8, MODE_TABLE, 31, n_bins=64, hist_lo=500.0, hist_hi=1500.0)
    r = eng.simulate(sim, want_final=False, want_stats=True)
    st = eng.read_stats(r.stats_raw)
    assert st.count == n and int(st.hist.sum()) + st.underflow + st.overflow == n
    # the last 1000 paths of that range, simulated on their own, agree with the oracle
    tail = eng.simulate(Engine.make_sim(1000, 8, MODE_TABLE, 31, first_path=n - 1000)).final.cpu().numpy()
    o = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 8, 1000, 31, first_path=n - 1000, table=table))
    assert np.array_equal(tail.view(np.uint32), o["final"].view(np.uint32))
    # mean of 8 periods of the table's mean return, within 5 standard errors
    m = float(np.mean(1.0 + table.astype(np.float64) / 100.0))
    sd = float(np.std(np.log1p(table.astype(np.float64) / 100.0))) * np.sqrt(8) * 1000 * m ** 8
    assert st.mean == pytest.approx(1000.0 * m ** 8, abs=5 * sd / np.sqrt(n) + 0.05)


@pytest.mark.parametrize("mode_name", ["table", "gaussian"])
def test_keepdata_full_size_properties(eng, oracle, table, mode_name, monkeypatch):
    """keepdata at benchmark size (4e6 paths x 360 periods, 5.8 GB on the device): properties that
    need no oracle run over everything, the oracle checks sampled rows."""
    import torch
    from stock_market_monte_carlo_amd import Engine
    mode = _modes()[mode_name]
    n, p, first = 4_000_000, 360, 123_456_789_000  # path ids past 2^32
    sim = Engine.make_sim(n, p, mode, SEED, first_path=first)
    traj, final = eng.simulate_keepdata(sim)
    assert bool((traj[:, 0] == 1000.0).all())                                   # values[0] = initial capital
    assert torch.equal(traj[:, p].contiguous().view(torch.int32), final.view(torch.int32))  # last column = final value
    # the same bits as the kernel that keeps only the final values
    only_final = eng.simulate(sim).final
    assert torch.equal(only_final.view(torch.int32), final.view(torch.int32))
    checksum = int(traj.view(torch.int32).to(torch.int64).sum().item())
    # sampled rows against the oracle's trajectory of that path (first, last, wave and chunk edges, random)
    rng = np.random.default_rng(5)
    rows = sorted({0, 1, 63, 64, 255, 256, n - 1, n - 64, n - 65, *rng.integers(0, n, 40).tolist()})
    op = oracle.make_params(mode, p, n, SEED, first_path=first, table=table)
    got = traj[torch.tensor(rows, device=traj.device)].cpu().numpy()
    for k, r in enumerate(rows):
        want = oracle.many_updates(1000.0, oracle.counter_path_returns(op, first + r), p)
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), r
    del traj, only_final
    # tuning knobs: same checksum of all bits
    monkeypatch.setenv("SMMC_KEEPDATA_WAVES", "8")
    monkeypatch.setenv("SMMC_KEEPDATA_TILE", "16")
    traj2, final2 = eng.simulate_keepdata(sim)
    assert int(traj2.view(torch.int32).to(torch.int64).sum().item()) == checksum
    assert torch.equal(final2.view(torch.int32), final.view(torch.int32))


def test_distribution_matches_reference_cpu_engine(eng, oracle, table):
    """Distribution-level parity with the reference CPU algorithm (engine R: mt19937 +
    Lemire + update_fund, src/simulations.cpp:240-252).  Different generators, same law:
    compare log-return moments within 5 standard errors."""
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE
    n, p = 200_000, 360
    ref, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, 424242)
    got = eng.simulate(Engine.make_sim(n, p, MODE_TABLE, 777)).final.cpu().numpy()
    lr, lg = np.log(ref.astype(np.float64)), np.log(got.astype(np.float64))
    se = np.sqrt(lr.var() / n + lg.var() / n)
    assert abs(lr.mean() - lg.mean()) < 5 * se
    assert abs(lr.std() - lg.std()) < 5 * lr.std() / np.sqrt(2 * n) * np.sqrt(2) * 1.5
    for q in (0.01, 0.25, 0.5, 0.75, 0.99):
        a, b = np.quantile(lr, q), np.quantile(lg, q)
        assert abs(a - b) < 0.03, (q, a, b)


def test_reference_named_api(table):
    """The reference's function names and error behaviour (simulations.h)."""
    import stock_market_monte_carlo_amd as S
    tot = S.mc_simulations_gpu(10000, 36, 1000.0, table, n_gpus=1, seed=5)
    assert tot.shape == (10000,) and tot.dtype == np.float32
    again = S.mc_simulations_gpu(10000, 36, 1000.0, table, n_gpus=1, seed=5)
    assert np.array_equal(tot.view(np.uint32), again.view(np.uint32))
    pre = np.full(10000, 1000.0, dtype=np.float32)  # caller pre-sizes (benchmark_mc_cpu_v2.cpp:26)
    ret = S.mc_simulations(10000, 36, 1000.0, table, final_values=pre, seed=5)
    assert np.array_equal(pre.view(np.uint32), tot.view(np.uint32)) and ret.base is pre or ret is pre
    means, variances = S.mc_simulations_gpu_reduceBlock(10000, 36, 1000.0, table, n_gpus=1, seed=5)
    assert means.shape == variances.shape == (40,)  # ceil(10000 / 256), src/simulations.cu:429-432
    np.testing.assert_allclose(means[0], tot[:256].astype(np.float64).mean(), rtol=1e-6)
    with pytest.raises(ValueError):  # src/simulations.cu:693
        S.mc_simulations_gpu_reduceBlock(1000, 36, 1000.0, table, n_gpus=2)
    data, fin = S.mc_simulations_keepdata(500, 36, 1000.0, table, seed=5)
    assert data.shape == (500, 37) and np.array_equal(data[:, -1], fin) and np.all(data[:, 0] == 1000.0)
    assert np.array_equal(fin.view(np.uint32), tot[:500].view(np.uint32))


def test_bad_arguments_are_errors_not_crashes(eng):
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE, SmmcError
    with pytest.raises(SmmcError):
        eng.simulate(Engine.make_sim(10, 10, 7, 1))  # unknown mode
    with pytest.raises(SmmcError):
        eng.simulate(Engine.make_sim(10, 10, MODE_TABLE, 1, n_bins=5000), want_stats=True)
    with pytest.raises(SmmcError):
        eng.simulate(Engine.make_sim(10, 10, MODE_TABLE, 1, n_bins=10, hist_lo=5, hist_hi=5), want_stats=True)
    with pytest.raises(SmmcError):
        eng.set_table(np.zeros(0, dtype=np.float32))
    with pytest.raises(SmmcError):
        eng.set_table(np.zeros(20000, dtype=np.float32))


def test_device_divide_shortcut_equals_ieee_divide(eng):
    """Exhaustive on-device check: the reciprocal-multiply divide by 100 over every normal
    binary32 with |x| >= 2^-90, both signs (3.6e9 patterns)."""
    import struct
    bits = lambda f: struct.unpack("<I", struct.pack("<f", f))[0]  # noqa: E731
    assert eng.selftest(bits(2.0 ** -90), 0x7F800000) == 0
    assert eng.selftest(0x80000000 | bits(2.0 ** -90), 0xFF800000) == 0
    assert eng.selftest(0, 1) == 0  # +0
