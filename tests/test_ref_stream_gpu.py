"""The reference CPU engine's OWN random stream on the device (SMMC_FLAG_STREAM_REF) against
  (1) tests/golden/libstdcxx_random.json `paths`: whole reference-style paths (src/simulations.cpp:
      240-252 with explicit seeds) computed by the SYSTEM libstdc++'s std::mt19937 and
      std::uniform_int_distribution<int> (oracle/pin/pin_libstdcxx.cpp) -- a fixture the C oracle did
      not produce;
  (2) oracle engine (R) (orc_ref_mc_simulations: hand-written mt19937 + Lemire map, pinned to the same
      fixture) at BASELINE configs[0] size (1e6 paths x 360 periods) and across both kernels' edges.
Final values bit for bit (binary32 patterns); bucket counts, below/underflow/overflow exact; sums 1e-12.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _engine(table, monkeypatch=None, **env):
    import stock_market_monte_carlo_amd as S
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    e = S.Engine(0)
    e.set_table(table)
    return e


@pytest.fixture(scope="module")
def eng(table):
    e = _engine(table)
    yield e
    e.close()


def _sim(n, p, seed, first=0, cap=1000.0, **kw):
    from stock_market_monte_carlo_amd import MODE_TABLE, Engine
    return Engine.make_sim(n, p, MODE_TABLE, seed, first_path=first, initial_capital=cap, stream="ref", **kw)


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_golden_paths_of_the_real_libstdcxx(eng, table):
    """HIP result == what std::mt19937 + std::uniform_int_distribution<int> + update_fund gave."""
    pin = json.load(open(os.path.join(HERE, "golden", "libstdcxx_random.json")))
    assert pin["table_len"] == table.size
    # 226 .. 228, 454 / 455, 623 .. 625, 681, 850, 908, 1077 / 1078, 1135, 1304, 1531, 1816 / 1817: where the device kernels
    # change how they obtain the generator's words (the windowed kernel's and the tree kernel's stretches, the tree
    # kernel's two instantiations, the generic kernel beyond)
    assert sorted({c["n_periods"] for c in pin["paths"]}) == [1, 4, 226, 227, 228, 360, 454, 455, 623, 624, 625, 681, 850, 908, 1000,
                                                              1077, 1078, 1135, 1304, 1531, 1816, 1817, 2000]
    for case in pin["paths"]:
        r = eng.simulate(_sim(32, case["n_periods"], case["seed0"], cap=case["initial_capital"]))
        eng.sync()
        got = [int(x) for x in _bits(r.final.cpu().numpy())]
        assert got == case["final_bits"], (case["n_periods"], case["seed0"])
        # the seed of path id is (uint32)(seed + id): the same paths as ids 7 .. 38 of seed0 - 7
        r = eng.simulate(_sim(32, case["n_periods"], case["seed0"] - 7, first=7, cap=case["initial_capital"]))
        eng.sync()
        assert [int(x) for x in _bits(r.final.cpu().numpy())] == case["final_bits"]


def test_paths_in_which_the_real_library_rejects_an_output(eng, table):
    """`rejecting_paths` of the fixture: seeds for which the system libstdc++'s uniform_int_distribution rejected
    a generator output on the way (found by counting engine calls: one more than periods).  On the device such
    a path is flagged by the windowed kernel and finished by the generic one (360 periods), or runs in the
    generic kernel from the start (1000): alone, and as lane 17 of a wave of ordinary paths."""
    pin = json.load(open(os.path.join(HERE, "golden", "libstdcxx_random.json")))
    cases = pin["rejecting_paths"]
    assert len(cases) == 12 and {c["n_periods"] for c in cases} == {360, 1000}
    for c in cases:
        assert c["engine_calls"] > c["n_periods"]
        r = eng.simulate(_sim(1, c["n_periods"], c["seed"], cap=c["initial_capital"]))
        eng.sync()
        assert int(_bits(r.final.cpu().numpy())[0]) == c["final_bits"], c
        r = eng.simulate(_sim(64, c["n_periods"], c["seed"] - 17, cap=c["initial_capital"]))
        eng.sync()
        assert int(_bits(r.final.cpu().numpy())[17]) == c["final_bits"], c


def test_baseline_config0_size_matches_oracle_engine_r(eng, oracle, table):
    """BASELINE configs[0]: 360 periods x 1e6 paths, every final value, the histogram and the counters."""
    n, p, seed0 = 1_000_000, 360, 1000
    want, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0)
    r = eng.simulate(_sim(n, p, seed0, n_bins=100, hist_lo=0.0, hist_hi=20000.0), want_final=True,
                     want_chunk_stats=True, want_stats=True)
    st = eng.read_stats(r.stats_raw)
    got = r.final.cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want))
    ost, ohist = oracle.values_stats(want, 1000.0, 100, 0.0, 20000.0)
    assert np.array_equal(st.hist, ohist)
    assert (st.count, st.below, st.underflow, st.overflow) == (n, ost.below, ost.underflow, ost.overflow)
    assert st.min == ost.min and st.max == ost.max
    assert st.sum == pytest.approx(ost.sum, rel=1e-12) and st.sumsq == pytest.approx(ost.sumsq, rel=1e-12)
    om, ov = oracle.chunk_mean_var(want)
    assert np.allclose(r.chunk_mean.cpu().numpy(), om, rtol=1e-6, atol=0)
    assert np.allclose(r.chunk_var.cpu().numpy(), ov, rtol=1e-5, atol=1e-3)
    # statistics only (no caller buffer for the final values): the same record
    r2 = eng.simulate(_sim(n, p, seed0, n_bins=100, hist_lo=0.0, hist_hi=20000.0), want_final=False, want_stats=True)
    st2 = eng.read_stats(r2.stats_raw)
    assert np.array_equal(st2.hist, st.hist) and st2.below == st.below and st2.sum == st.sum


# 226/227/228: the windowed kernel's second stretch begins at output 227; 454 is the last length it takes; from 455 to
# 1816 the tree kernel carries the recurrence on (its stretches begin at outputs 454, 623, 681, 850, 908, 1077, 1135,
# 1246, 1304, 1362, 1473, 1531, 1589, 1700, 1758: lengths on both sides of each, and lengths that leave one, two or
# three outputs to the remainder loop); from 1817 on the generic kernel runs (624/625: its state wraps; 1248: twice)
@pytest.mark.parametrize("p", [0, 1, 2, 7, 8, 9, 226, 227, 228, 229, 360, 437, 438, 439, 453, 454, 455, 456, 457, 458, 622, 623,
                               624, 625, 626, 680, 681, 682, 683, 849, 850, 851, 852, 907, 908, 909, 910, 911, 1000, 1076,
                               1077, 1078, 1079, 1134, 1135, 1136, 1245, 1246, 1247, 1248, 1300, 1303, 1304, 1305, 1361, 1362,
                               1363, 1472, 1473, 1474, 1530, 1531, 1532, 1588, 1589, 1590, 1699, 1700, 1701, 1757, 1758, 1759,
                               1814, 1815, 1816, 1817, 1818, 1900])
def test_every_length_both_kernels(eng, oracle, table, p):
    n, seed0 = 2000 + 77, 2 ** 32 - 1000  # ragged; the seeds seed0 + id wrap past 2^32 inside the launch (path 1000 has seed 0)
    want, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0)
    r = eng.simulate(_sim(n, p, seed0))
    eng.sync()
    assert np.array_equal(_bits(r.final.cpu().numpy()), _bits(want))


@pytest.mark.parametrize("kernel", ["windowed", "tree", "generic"])
@pytest.mark.parametrize("exact_div", [False, True])
def test_forced_kernels_and_divides_agree(table, oracle, monkeypatch, kernel, exact_div):
    """SMMC_REF_KERNEL: "tree" runs ref_tree_kernel also for the lengths ref_windowed_kernel takes by default (its
    first two stretches are the same recurrence), "generic" the state-in-memory kernel for every length."""
    e = _engine(table, monkeypatch, SMMC_REF_KERNEL=kernel)
    try:
        for p in (5, 300, 360, 700, 1400):
            n, seed0 = 30011, 99
            want, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0)
            r = e.simulate(_sim(n, p, seed0, exact_div=exact_div))
            e.sync()
            assert np.array_equal(_bits(r.final.cpu().numpy()), _bits(want)), (kernel, exact_div, p)
        # trajectories through the forced kernel too (the tree kernel's short instantiation at a windowed length)
        n, p, seed0 = 777, 300, 4242
        traj, fin = e.simulate_keepdata(_sim(n, p, seed0, exact_div=exact_div))
        e.sync()
        rows = traj.cpu().numpy()
        want, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0)
        assert np.array_equal(_bits(rows[:, -1]), _bits(want)) and np.array_equal(_bits(fin.cpu().numpy()), _bits(want))
        for i in (0, 255, 256, n - 1):
            idx = oracle.mt19937_indices((seed0 + i) & 0xFFFFFFFF, table.size, p)
            assert np.array_equal(_bits(rows[i]), _bits(oracle.many_updates(1000.0, table[idx], p))), (kernel, i)
    finally:
        e.close()


def test_rejections_take_the_redo_path(table, oracle, monkeypatch):
    """The windowed kernel only flags a path that rejects a generator output: it goes on the redo list and
    the generic kernel finishes it.  T = 12289 rejects 2.1e-6 of the outputs: a few hundred of the 4e5
    paths; every one of them must come back with the oracle's bits, whichever kernel is forced."""
    rng = np.random.default_rng(5)
    big = rng.normal(0.6, 4.3, 12289).astype(np.float32)
    threshold = (2 ** 32 - big.size) % big.size
    n, p, seed0 = 400_000, 360, 31337
    want, _ = oracle.ref_mc_simulations(n, p, 1000.0, big, seed0)
    expect_rejecting = n * (1.0 - (1.0 - threshold / 2.0 ** 32) ** p)  # paths with at least one rejected output
    assert 50 < expect_rejecting < 5000
    for kernel in ("auto", "windowed", "tree", "generic"):
        e = _engine(big, monkeypatch, SMMC_REF_KERNEL=kernel)
        try:
            r = e.simulate(_sim(n, p, seed0))
            e.sync()
            assert np.array_equal(_bits(r.final.cpu().numpy()), _bits(want)), kernel
        finally:
            e.close()
    # the tree kernel's own lengths: 1000 periods, where one path in 350 rejects an output somewhere
    n2, p2 = 100_000, 1000
    want2, _ = oracle.ref_mc_simulations(n2, p2, 1000.0, big, seed0)
    e = _engine(big, monkeypatch, SMMC_REF_KERNEL="auto")
    try:
        r = e.simulate(_sim(n2, p2, seed0))
        e.sync()
        assert np.array_equal(_bits(r.final.cpu().numpy()), _bits(want2))
    finally:
        e.close()


def test_checked_divide_window_and_overflowing_paths(table, oracle, monkeypatch):
    """A table with +60 % entries cannot be proven safe for the fast divide over 400 periods (CHECKED);
    with a capital of 1e30 most paths really leave the window and about half overflow to inf: those come
    back from the generic kernel with the IEEE divide, bit for bit the oracle's."""
    from stock_market_monte_carlo_amd._lib import DIV_CHECKED
    t = table.copy()
    t[::10] = 60.0
    e = _engine(t, monkeypatch)
    try:
        for cap, p in ((1000.0, 400), (1e30, 700), (1e30, 400)):  # 700 periods: the tree kernel's checked variant (all overflow)
            sim = _sim(20011, p, 7, cap=cap)
            assert e.divide_kind(sim) == DIV_CHECKED
            want, _ = oracle.ref_mc_simulations(20011, p, cap, t, 7)
            r = e.simulate(sim)
            e.sync()
            got = r.final.cpu().numpy()
            assert np.array_equal(_bits(got), _bits(want)), (cap, p)
        assert np.isinf(want).any() and np.isfinite(want).any()
    finally:
        e.close()


def test_host_pipeline_chunks_and_shards(eng, oracle, table, monkeypatch):
    """simulate_to_host in several chunks, and two shards of one run == the whole run (first_path offsets)."""
    n, p, seed0 = 300_000, 360, 123456789
    want, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0)
    e = _engine(table, monkeypatch, SMMC_HOST_CHUNK_PATHS=65536)
    try:
        host, st, _ = e.simulate_to_host(_sim(n, p, seed0, n_bins=50, hist_lo=0.0, hist_hi=30000.0), want_stats=True)
        assert np.array_equal(_bits(host), _bits(want))
        ost, ohist = oracle.values_stats(want, 1000.0, 50, 0.0, 30000.0)
        assert np.array_equal(st.hist, ohist) and st.below == ost.below and st.count == n
    finally:
        e.close()
    a = eng.simulate(_sim(100_001, p, seed0))
    b = eng.simulate(_sim(n - 100_001, p, seed0, first=100_001))
    eng.sync()
    assert np.array_equal(_bits(np.concatenate([a.final.cpu().numpy(), b.final.cpu().numpy()])), _bits(want))


def test_argument_errors(eng):
    from stock_market_monte_carlo_amd import MODE_GAUSSIAN, Engine, SmmcError
    sim = Engine.make_sim(10, 4, MODE_GAUSSIAN, 1, stream="ref")
    with pytest.raises(SmmcError, match="table mode only"):
        eng.simulate(sim)
    # empty launch: a zero record, nothing written
    r = eng.simulate(_sim(0, 360, 1, n_bins=10, hist_lo=0.0, hist_hi=1.0), want_final=True, want_stats=True)
    st = eng.read_stats(r.stats_raw)
    assert st.count == 0 and int(st.hist.sum()) == 0


@pytest.mark.parametrize("p", [0, 1, 31, 32, 33, 64, 227, 360, 454, 455, 700, 1000, 1077, 1078, 1500, 1816, 1817])
def test_trajectories_of_the_reference_stream(eng, oracle, table, p):
    """mc_simulations_keepdata draws like mc_simulations (src/simulations.cpp:175-186: a generator per path,
    sample_returns_historical, many_updates): with SMMC_FLAG_STREAM_REF every row is many_updates of the table
    entries the path's own mt19937 + Lemire map picks -- checked value by value against the oracle's generator."""
    n, seed0 = 700 + 13, 2 ** 32 - 300  # the seeds wrap past 2^32 at path 300
    traj, final = eng.simulate_keepdata(_sim(n, p, seed0))
    eng.sync()
    got = traj.cpu().numpy()
    want_final, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0)
    assert got.shape == (n, p + 1) and np.all(got[:, 0] == np.float32(1000.0))
    assert np.array_equal(_bits(got[:, -1]), _bits(want_final)) and np.array_equal(_bits(final.cpu().numpy()), _bits(want_final))
    for i in (0, 1, 63, 64, 255, 256, 299, 300, n - 1):  # rows across waves, workgroups and the seed wrap, value by value
        idx = oracle.mt19937_indices((seed0 + i) & 0xFFFFFFFF, table.size, p)
        row = oracle.many_updates(1000.0, table[idx], p)
        assert np.array_equal(_bits(got[i]), _bits(row)), (p, i)


def test_trajectories_of_the_real_libstdcxx(eng, table):
    """The fixture's `trajectories`: sample_returns_historical + many_updates as mc_simulations_keepdata runs them
    (src/simulations.cpp:95-112, 175-186), computed by the system libstdc++ -- every value of the row, incl. a path
    that rejects an output (seed 32569) and the largest 32-bit seed."""
    pin = json.load(open(os.path.join(HERE, "golden", "libstdcxx_random.json")))
    assert len(pin["trajectories"]) == 9 and {c["n_periods"] for c in pin["trajectories"]} == {40, 360, 1000}  # 1000: the tree kernel
    for c in pin["trajectories"]:
        traj, final = eng.simulate_keepdata(_sim(3, c["n_periods"], c["seed"] - 1, cap=c["initial_capital"]))
        eng.sync()
        row = traj.cpu().numpy()[1]  # path id 1 of seed - 1
        assert [int(x) for x in _bits(row)] == c["value_bits"], (c["n_periods"], c["seed"])
        assert int(_bits(final.cpu().numpy())[1]) == c["value_bits"][-1]


def test_trajectories_through_the_redo_path_and_to_host(oracle, monkeypatch):
    """Rows of paths that reject a generator output come from the generic kernel (T = 12289: a few hundred of
    4e5 paths): every row's last value is the oracle's final value, and rows are whole (first value the capital,
    a row equal to many_updates of its own draws for sampled paths incl. every rejecting one found)."""
    rng = np.random.default_rng(5)
    big = rng.normal(0.6, 4.3, 12289).astype(np.float32)
    n, p, seed0 = 400_000, 360, 31337
    want, _ = oracle.ref_mc_simulations(n, p, 1000.0, big, seed0)
    for kernel in ("auto", "generic"):
        e = _engine(big, monkeypatch, SMMC_REF_KERNEL=kernel)
        try:
            traj, final = e.simulate_keepdata_to_host(_sim(n, p, seed0))
            assert np.array_equal(_bits(final), _bits(want)) and np.array_equal(_bits(traj[:, -1]), _bits(want)), kernel
            assert np.all(traj[:, 0] == np.float32(1000.0))
            threshold = (2 ** 32 - big.size) % big.size
            checked = rejecting = 0
            for i in range(0, n, 401):
                raw = oracle.mt19937_raw((seed0 + i) & 0xFFFFFFFF, p + 8)
                low = (raw.astype(np.uint64) * big.size) & 0xFFFFFFFF
                rejecting += int((low[:p] < threshold).any())
                idx = oracle.mt19937_indices((seed0 + i) & 0xFFFFFFFF, big.size, p)
                assert np.array_equal(_bits(traj[i]), _bits(oracle.many_updates(1000.0, big[idx], p))), (kernel, i)
                checked += 1
            assert checked > 900
        finally:
            e.close()


def test_full_size_properties_of_the_reference_stream(eng, oracle, table):
    """BASELINE configs[2] size on the reference's own stream (1e8 paths x 360 periods, 400 MB of final values):
    what does not need the oracle at that size -- count conservation, statistics against a torch reduction of
    the final values, bit determinism from run to run, two shards equal to the whole -- and the oracle on the
    first and last 300 paths and on 300 paths around the 2^27-path launch boundary's place in a longer run."""
    import torch
    n, p, seed0 = 100_000_000, 360, 4242
    r = eng.simulate(_sim(n, p, seed0, n_bins=100, hist_lo=0.0, hist_hi=20000.0), want_stats=True)
    st = eng.read_stats(r.stats_raw)
    f = r.final
    assert st.count == n and int(st.hist.sum()) + st.underflow + st.overflow == n
    assert st.sum == pytest.approx(float(f.double().sum()), rel=1e-12)
    assert st.below == int((f < 1000.0).sum()) and st.min == float(f.min()) and st.max == float(f.max())
    again = eng.simulate(_sim(n, p, seed0)).final
    assert torch.equal(f.view(torch.int32), again.view(torch.int32))
    half = 50_000_017
    a = eng.simulate(_sim(half, p, seed0)).final
    b = eng.simulate(_sim(n - half, p, seed0, first=half)).final
    assert torch.equal(torch.cat([a, b]).view(torch.int32), f.view(torch.int32))
    host = f.cpu().numpy()
    for first in (0, half - 150, n - 300):
        want, _ = oracle.ref_mc_simulations(300, p, 1000.0, table, seed0 + first)
        assert np.array_equal(_bits(host[first:first + 300]), _bits(want)), first
    # more than one launch per call (2^27 paths each): a short-period run keeps it quick
    n2 = (1 << 27) + 1000
    g = eng.simulate(_sim(n2, 2, seed0)).final.cpu().numpy()
    for first in (0, (1 << 27) - 150, n2 - 300):
        want, _ = oracle.ref_mc_simulations(300, 2, 1000.0, table, seed0 + first)
        assert np.array_equal(_bits(g[first:first + 300]), _bits(want)), first


@pytest.mark.parametrize("n,p,shift", [(3 * 2048 + 77, 360, 0), (2 * 2048 + 1, 361, 3), (5000, 63, 17), (4096, 1000, 1),
                                       (2049, 32, 31), (10000, 5, 7), (2048, 31, 9), (6000, 0, 5)])
def test_comb_trajectories_every_row_and_nothing_else(eng, oracle, table, n, p, shift):
    """Round 4's trajectory writer (TrajWriter, csrc/smmc_ref_kernels.hip): a lane runs eight consecutive rows,
    lanes are 32 rows apart, four waves share a 2048-row super-chunk, every 128-byte line leaves as a whole.
    Several super-chunks with a ragged last one, row lengths that are and are not multiples of a line, base
    pointers at every kind of offset inside a line: EVERY row equals many_updates of its own generator's draws,
    and the floats before and after the array keep their sentinel."""
    import ctypes as C
    import torch
    from stock_market_monte_carlo_amd import _lib
    seed0, guard = 77_000 + n, 64
    row = p + 1
    buf = torch.full((guard + shift + n * row + guard,), float("nan"), dtype=torch.float32, device=eng.tdevice)
    fin = torch.full((n + 2,), float("nan"), dtype=torch.float32, device=eng.tdevice)
    traj = buf[guard + shift:guard + shift + n * row]
    sim = _sim(n, p, seed0)
    eng._enter()
    _lib.check(eng._L.smmc_engine_simulate_keepdata(eng._h, C.byref(sim), C.c_void_p(traj.data_ptr()), C.c_void_p(fin[1:].data_ptr())))
    eng.sync()
    host = buf.cpu().numpy()
    assert np.isnan(host[:guard + shift]).all() and np.isnan(host[guard + shift + n * row:]).all()
    f = fin.cpu().numpy()
    assert np.isnan(f[0]) and np.isnan(f[-1])
    got = host[guard + shift:guard + shift + n * row].reshape(n, row)
    want_final, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0)
    assert np.array_equal(_bits(f[1:-1]), _bits(want_final)) and np.array_equal(_bits(got[:, -1]), _bits(want_final))
    assert np.all(got[:, 0] == np.float32(1000.0))
    step = 1 if n * row <= 1_500_000 else 7  # every row of the smaller shapes, every seventh (+ the edges) of the larger
    rows = sorted(set(range(0, n, step)) | set(range(0, 80)) | set(range(2040, min(n, 2060))) | set(range(max(0, n - 70), n)))
    for i in rows:
        idx = oracle.mt19937_indices((seed0 + i) & 0xFFFFFFFF, table.size, p)
        want = oracle.many_updates(1000.0, table[idx], p) if p else np.array([1000.0], dtype=np.float32)
        assert np.array_equal(_bits(got[i]), _bits(want)), (n, p, shift, i)


@pytest.mark.parametrize("rows", [1, 2, 4, 8])
@pytest.mark.parametrize("p", [360, 1000])
def test_comb_trajectories_do_not_depend_on_the_rows_per_stream(eng, oracle, table, monkeypatch, rows, p):
    """SMMC_REF_TRAJ_ROWS (the consecutive rows a lane runs: 8 / K workgroups share a 2048-row super-chunk): the same
    trajectories for every setting, rows across streams, waves, workgroups and the ragged last super-chunk against
    the oracle (windowed kernel at 360 periods, tree kernel at 1000)."""
    monkeypatch.setenv("SMMC_REF_TRAJ_ROWS", str(rows))
    n, seed0 = 2 * 2048 + 333, 5_000_000
    traj, final = eng.simulate_keepdata(_sim(n, p, seed0))
    eng.sync()
    got = traj.cpu().numpy()
    want_final, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0)
    assert np.array_equal(_bits(got[:, -1]), _bits(want_final)) and np.array_equal(_bits(final.cpu().numpy()), _bits(want_final))
    for i in list(range(0, 70)) + list(range(2040, 2120)) + list(range(4090, n)) + list(range(500, 4000, 97)):
        idx = oracle.mt19937_indices((seed0 + i) & 0xFFFFFFFF, table.size, p)
        assert np.array_equal(_bits(got[i]), _bits(oracle.many_updates(1000.0, table[idx], p))), (rows, p, i)
