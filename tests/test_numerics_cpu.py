"""CPU proofs behind the device-side arithmetic shortcuts, using the
oracle's scanners: the shortcuts themselves are re-stated there with fmaf()."""
import struct

import numpy as np
import pytest


def bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def test_div100_shortcut_is_exact_on_its_whole_domain(oracle):
    """fma(x, ch, fl(x cl)) with ch = fl(1/100), cl = fl(1/100 - ch) equals x / 100.0f for EVERY binary32
    with |x| >= 2^-114 (all positive patterns; negatives are symmetric and covered by the device
    self-test).  It fails only where x cl loses bits to gradual underflow; the host uses it from 2^-90
    up, where x cl is normal."""
    import ctypes as C
    L = oracle.lib()
    L.orc_div100_cl.restype = C.c_float
    assert L.orc_div100_cl() == float.fromhex("0x1.eb851ep-33")
    bad, first = oracle.div100_mismatches(bits(2.0 ** -114), 0x7F800000)
    assert bad == 0, hex(first)
    assert oracle.div100_mismatches(0, 1)[0] == 0  # +0
    # and the documented failure region really exists (so the host-side guard is needed)
    assert oracle.div100_mismatches(bits(2.0 ** -126), bits(2.0 ** -114))[0] > 0


def test_box_muller_radius_table_accuracy(oracle):
    """r = sqrt(-2 ln U) from the piecewise-cubic table, U taken from fl(2 w + 1) / 2^33 (the
    distance from the nearer end, rounded to binary32): within one binary32 ulp of r (4.8e-7 at
    the 6.76-sigma end) over a dense sweep of all octaves."""
    assert oracle.bm_radius_scan(0, 2 ** 32, 499) < 6e-7
    assert oracle.bm_radius_scan(0, 1 << 20, 1) < 6e-7                   # deepest tail, every value
    assert oracle.bm_radius_scan((1 << 32) - (1 << 20), 1 << 32, 1) < 6e-7  # U -> 1 end (sqrt singularity)
    assert oracle.bm_radius_scan((1 << 31) - (1 << 18), (1 << 31) + (1 << 18), 1) < 6e-7  # where the sides meet
    assert oracle.bm_radius(0) == max(oracle.bm_radius(a) for a in (0, 1, 2, 1000, 2 ** 31))
    assert abs(oracle.bm_radius(0) - 6.7637) < 1e-3 and 0 < oracle.bm_radius(0xFFFFFFFF) < 2e-5


def test_box_muller_tables_identical_in_oracle_and_product():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a = open(os.path.join(root, "oracle", "smmc_bm_tables.inc")).read()
    b = open(os.path.join(root, "stock_market_monte_carlo_amd", "csrc", "smmc_bm_tables.inc")).read()
    assert a == b and "SMMC_BM_RADIUS_ENTRIES 1056" in a and "SMMC_BM3_RADIUS_ENTRIES 512" in a and "SMMC_BM3_TRIG_ENTRIES 2048" in a


def test_box_muller_moments_and_accuracy(oracle):
    rng = np.random.default_rng(5)
    ua = rng.integers(0, 2 ** 32, 100000, dtype=np.uint64)
    ub = rng.integers(0, 2 ** 32, 100000, dtype=np.uint64)
    z = np.array([oracle.box_muller(int(a), int(b)) for a, b in zip(ua, ub)])
    r = np.sqrt(-2 * np.log((2 * ua.astype(np.float64) + 1) / 2.0 ** 33))
    th = 2 * np.pi * ub.astype(np.float64) / 2.0 ** 32
    assert np.abs(z[:, 0] - r * np.cos(th)).max() < 2e-6
    assert np.abs(z[:, 1] - r * np.sin(th)).max() < 2e-6
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    # extremes of the input words stay finite and bounded
    # the scaled form the kernels use: fma(r * scale, cos / sin, shift)
    zc, zs = oracle.box_muller(123456789, 987654321)
    dc, ds = oracle.box_muller_scaled(123456789, 987654321, 0.83333, 0.5)
    assert dc == pytest.approx(0.5 + 0.83333 * zc, abs=2e-6) and ds == pytest.approx(0.5 + 0.83333 * zs, abs=2e-6)
    for a, b in [(0, 0), (0xFFFFFFFF, 0xFFFFFFFF), (0, 0x80000000), (0xFFFFFFFF, 0x40000000), (1, 0x3FFFFFFF),
                 (0x7FFFFFFF, 0x007FFFFF), (0x80000000, 0x00800000), (0x12345678, 0xFF800000)]:
        zc, zs = oracle.box_muller(a, b)
        assert np.isfinite(zc) and np.isfinite(zs) and abs(zc) < 7 and abs(zs) < 7
        assert abs(np.hypot(zc, zs) - oracle.bm_radius(a)) < 1e-5 * (1 + oracle.bm_radius(a))
    # angle 0 and the quarter turns come out exact
    assert oracle.box_muller(0, 0) == (oracle.bm_radius(0), 0.0)
    zc, zs = oracle.box_muller(0, 0x40000000)
    assert zc == 0.0 and zs == oracle.bm_radius(0)


def _v3_u(ua):
    """The uniform's distance from the nearer end of (0, 1) as counter stream v3 defines it, and its side."""
    d = np.asarray(ua, dtype=np.int64)
    d = np.where(d >= 2 ** 31, d - 2 ** 32, d)                       # the word read as int32
    f = d.astype(np.float32).astype(np.float64)                      # rounded to binary32
    return np.where(d == 0, 2.0 ** -33, np.abs(f) / 2.0 ** 32), d < 0


def test_stream_v3_radius_table_accuracy(oracle):
    """Counter stream v3's radius: f = fl(d), d the word read as int32 (the signed distance from the
    nearer end), u = |f| / 2^32 in (0, 1/2] (d = 0: 2^-33), 32 octaves of 8 sub-intervals stored rotated so
    that the bin is one bit-field of f's pattern and the side its sign, cubic in f itself.  Same bound as
    v2's table."""
    assert oracle.bm3_radius_scan(0, 2 ** 32, 499) < 6e-7
    assert oracle.bm3_radius_scan(0, 1 << 20, 1) < 6e-7                       # deepest tail, every value
    assert oracle.bm3_radius_scan((1 << 32) - (1 << 20), 1 << 32, 1) < 6e-7  # U -> 1 end (sqrt singularity)
    assert oracle.bm3_radius_scan((1 << 31) - (1 << 18), (1 << 31) + (1 << 18), 1) < 6e-7  # where the sides meet
    # octave boundaries: every power of two of the distance, both neighbours, both sides
    edges = np.array([v for e in range(1, 31) for v in ((1 << e) - 1, 1 << e, (1 << e) + 1)], dtype=np.int64)
    for side in (0, 1):
        words = edges if side == 0 else (1 << 32) - edges
        got = np.array([oracle.bm3_radius(int(w)) for w in words])
        u, neg = _v3_u(words)
        assert np.all(neg == bool(side))
        want = np.sqrt(-2 * np.log1p(-u)) if side else np.sqrt(-2 * np.log(u))
        assert np.abs(got - want).max() < 6e-7
    # the word 0 stands for u = 2^-33 and is the largest radius; then 1 / 2^32, 2 / 2^32, ...
    assert oracle.bm3_radius(0) == pytest.approx(np.sqrt(-2 * np.log(2.0 ** -33)), abs=1e-6)      # 6.76 sigma
    assert oracle.bm3_radius(1) == pytest.approx(np.sqrt(-2 * np.log(2.0 ** -32)), abs=1e-6)      # 6.66 sigma
    assert oracle.bm3_radius(0) > oracle.bm3_radius(1) > oracle.bm3_radius(2) > oracle.bm3_radius(3) > oracle.bm3_radius(1000)
    assert 0 < oracle.bm3_radius(0xFFFFFFFF) < 3e-5
    # u = 1/2 from either side: the word 2^31 (d = -2^31) and the words that round to +2^31
    assert oracle.bm3_radius(0x80000000) == oracle.bm3_radius(0x7FFFFFFF) == oracle.bm3_radius(0x7FFFFFC0)
    assert oracle.bm3_radius(0x80000000) == pytest.approx(np.sqrt(2 * np.log(2.0)), abs=3e-7)
    # the scaled form multiplies the coefficients, not the result
    assert oracle.lib().orc_bm3_radius_scaled is not None


def test_stream_v3_box_muller_moments_and_accuracy(oracle):
    """v3's angle is the low 30 bits of the second word, theta = 2 pi (ub mod 2^30) / 2^30, and the
    table's (cos, sin) are rotated by the residual angle to FIRST order: the draw is r cos(theta) times
    sqrt(1 + delta^2) kappa, a factor within -3.9e-7 .. +7.8e-7 of 1 whose mean square is 1, at an
    angle off by delta^3/3 <= 1.2e-9.  Checked here: against the exact r cos / sin relative to 1 + r,
    the length factor's range and mean square, and the moments."""
    rng = np.random.default_rng(5)
    ua = rng.integers(0, 2 ** 32, 100000, dtype=np.uint64)
    ub = rng.integers(0, 2 ** 32, 100000, dtype=np.uint64)
    z = np.array([oracle.box_muller3(int(a), int(b)) for a, b in zip(ua, ub)])
    u, neg = _v3_u(ua)
    r = np.where(neg, np.sqrt(-2 * np.log1p(-u)), np.sqrt(-2 * np.log(u)))
    low = (ub % np.uint64(2 ** 30)).astype(np.float64)
    th = 2 * np.pi * low / 2.0 ** 30
    # |dz|: radius table 5.5e-7 + length factor 7.8e-7 r + binary32 roundings
    assert (np.abs(z[:, 0] - r * np.cos(th)) / (1 + r)).max() < 1.0e-6
    assert (np.abs(z[:, 1] - r * np.sin(th)) / (1 + r)).max() < 1.0e-6
    assert np.abs(z[:, 0] - r * np.cos(th)).max() < 5e-6 and np.abs(z[:, 1] - r * np.sin(th)).max() < 5e-6
    # without the length factor (the first-order rotation evaluated in double) what is left is the
    # radius table and roundings: the same 1.2e-6 as a second-order rotation gave
    n_sec = 2048
    dmax = np.pi / n_sec
    delta = th - 2 * np.pi * (np.floor(low / 2.0 ** 19) + 0.5) / n_sec
    assert np.abs(delta).max() <= dmax * (1 + 1e-12)
    factor = np.sqrt(1 + delta ** 2) / np.sqrt(1 + dmax ** 2 / 3)
    th1 = th - delta + np.arctan(delta)
    assert np.abs(z[:, 0] - factor * r * np.cos(th1)).max() < 1.2e-6
    assert np.abs(z[:, 1] - factor * r * np.sin(th1)).max() < 1.2e-6
    assert factor.min() > 1 - 3.95e-7 and factor.max() < 1 + 7.9e-7 and abs((factor ** 2).mean() - 1) < 2e-8
    big = r > 0.5
    length = np.hypot(z[big, 0], z[big, 1]) / r[big]
    assert abs((length ** 2).mean() - 1) < 5e-8          # no net scale: the variance of the draws is kept
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    # the word's top two bits do not enter the angle
    assert oracle.box_muller3(12345, 0x12345678) == oracle.box_muller3(12345, 0xD2345678)
    # the scaled form: the kernels draw the MULTIPLIER fma(r std, cos, 100 + mean)
    zc, zs = oracle.box_muller3(123456789, 987654321)
    dc, ds = oracle.box_muller3_scaled(123456789, 987654321, 0.83333, 100.5)
    assert dc == pytest.approx(100.5 + 0.83333 * zc, abs=1e-5) and ds == pytest.approx(100.5 + 0.83333 * zs, abs=1e-5)
    for a, b in [(0, 0), (0xFFFFFFFF, 0xFFFFFFFF), (0, 0x80000000), (0xFFFFFFFF, 0x40000000), (1, 0x3FFFFFFF),
                 (0x7FFFFFFF, 0x0007FFFF), (0x80000000, 0x00080000), (0x12345678, 0xFFF80000), (5, 0xFFFFFFFF)]:
        zc, zs = oracle.box_muller3(a, b)
        assert np.isfinite(zc) and np.isfinite(zs) and abs(zc) < 7 and abs(zs) < 7
        assert abs(np.hypot(zc, zs) - oracle.bm3_radius(a)) < 1e-5 * (1 + oracle.bm3_radius(a))
    # the table angles are sector middles, so nothing is exact at the axes, only within rounding and the
    # length factor; at the middle of a sector delta is 0 to within 1e-9 and the table entry comes back
    r0 = oracle.bm3_radius(0)
    zc, zs = oracle.box_muller3(0, 0)
    assert zc == pytest.approx(r0, rel=1e-6) and abs(zs) < 1e-5
    zc, zs = oracle.box_muller3(0, 0x10000000)           # a quarter of 2^30
    assert abs(zc) < 1e-5 and zs == pytest.approx(r0, rel=1e-6)
    zc, zs = oracle.box_muller3(0, 0x3FFFFFFF)           # 2 pi - epsilon: last sector, upper edge
    assert zc == pytest.approx(r0, rel=1e-6) and abs(zs) < 1e-5
    zc, zs = oracle.box_muller3(0, (37 << 19) | (1 << 18))
    kappa = 1 / np.sqrt(1 + dmax ** 2 / 3)
    assert zc == pytest.approx(r0 * kappa * np.cos(2 * np.pi * 37.5 / n_sec), rel=3e-7)
    assert zs == pytest.approx(r0 * kappa * np.sin(2 * np.pi * 37.5 / n_sec), rel=3e-7)


def test_stream_v3_return_is_the_multiplier_minus_100(oracle, table):
    """v3 draws the multiplier a; the period return it reports is a - 100, exact by Sterbenz for
    a in [50, 200], so update_fund(total, return) reproduces the engine's step bit for bit: the
    trajectory IS many_updates of the reported returns (src/simulations.cpp:18-22)."""
    p = oracle.make_params(oracle.MODE_GAUSSIAN, 360, 4, 99, first_path=12345)
    r = oracle.counter_mc(p, want_traj=True)
    for i in range(4):
        rets = oracle.counter_path_returns(p, 12345 + i)
        assert np.array_equal(oracle.many_updates(1000.0, rets, 360).view(np.uint32), r["traj"][i].view(np.uint32))
    # the two streams are different transforms of different Philox words (v3 counts blocks in the
    # counter's first word, v2 in its third): other paths, the same distribution
    p2 = oracle.make_params(oracle.MODE_GAUSSIAN, 360, 4, 99, first_path=12345, stream=2)
    r2 = oracle.counter_path_returns(p2, 12345).astype(np.float64)
    r3 = oracle.counter_path_returns(p, 12345).astype(np.float64)
    assert not np.array_equal(r2, r3)
    assert abs(r2.mean() - r3.mean()) < 5 * 0.83333 * np.sqrt(2.0 / 360)
    # table mode: the same draw from other Philox words (v3 counts blocks in the counter's first word,
    # v2 in its third): different paths, the same distribution
    t3 = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 72, 4000, 7, table=table))["final"]
    t2 = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 72, 4000, 7, table=table, stream=2))["final"]
    assert not np.array_equal(t2.view(np.uint32), t3.view(np.uint32))
    assert abs(np.log(t2).mean() - np.log(t3).mean()) < 5 * np.log(t2).std() * np.sqrt(2.0 / 4000)


def test_histogram_bucket_contract(oracle):
    L = oracle.lib()
    assert L.orc_hist_bucket(-1.0, 0.0, 100.0, 10) == -1
    assert L.orc_hist_bucket(0.0, 0.0, 100.0, 10) == 0
    assert L.orc_hist_bucket(9.999999, 0.0, 100.0, 10) == 0
    assert L.orc_hist_bucket(10.0, 0.0, 100.0, 10) == 1
    assert L.orc_hist_bucket(99.99999, 0.0, 100.0, 10) == 9
    assert L.orc_hist_bucket(100.0, 0.0, 100.0, 10) == 10
    assert L.orc_hist_bucket(float("nan"), 0.0, 100.0, 10) == 10
    assert L.orc_hist_bucket(float("inf"), 0.0, 100.0, 10) == 10


def test_dense_table_digits_match_big_integer_arithmetic(oracle, table):
    """Tables <= 2048 entries: eight indices per Philox block by base-T digit extraction from the
    two 64-bit halves.  Checked against exact Python integers, with the counter of either stream
    (v3: (block, path_lo, path_hi, mode); v2: (path_lo, path_hi, block, mode))."""
    seed = 0xABCDEF0123456789
    for T in (1127, 1, 2, 2048, 1000):
        tab = np.arange(T, dtype=np.float32)  # entry value == index
        for stream in (3, 2):
            p = oracle.make_params(oracle.MODE_TABLE, 24, 1, seed, table=tab, stream=stream)
            for path in (0, 5, (1 << 33) + 17):
                got = oracle.counter_path_indices(p, path)
                want = []
                for blk in range(3):
                    lo32, hi32 = path & 0xFFFFFFFF, path >> 32
                    ctr = [blk, lo32, hi32, 0] if stream == 3 else [lo32, hi32, blk, 0]
                    u = [int(x) for x in oracle.philox4x32_10(ctr, [seed & 0xFFFFFFFF, seed >> 32])]
                    for hi, lo in ((u[0], u[1]), (u[2], u[3])):
                        x = (hi << 32) | lo
                        for _ in range(3):
                            prod = x * T
                            want.append(prod >> 64)
                            x = prod & ((1 << 64) - 1)
                        want.append(((x >> 32) * T) >> 32)
                assert [int(v) for v in got] == want, (T, path, stream)
                # and the draws are the table entries at those indices
                assert np.array_equal(oracle.counter_path_returns(p, path), tab[got])


def test_sparse_schedule_for_large_tables(oracle):
    """Tables above 2048 entries: one index per 32-bit word, (u * T) >> 32."""
    T, seed = 5000, 99
    tab = np.arange(T, dtype=np.float32)
    p = oracle.make_params(oracle.MODE_TABLE, 8, 1, seed, table=tab)
    got = oracle.counter_path_indices(p, 3)
    want = []
    for blk in range(2):
        u = oracle.philox4x32_10([blk, 3, 0, 0], [seed, 0])
        want += [(int(x) * T) >> 32 for x in u]
    assert [int(v) for v in got] == want
    assert oracle.lib().orc_draws_per_block(oracle.MODE_TABLE, 2048) == 8
    assert oracle.lib().orc_draws_per_block(oracle.MODE_TABLE, 2049) == 4
    assert oracle.lib().orc_draws_per_block(oracle.MODE_GAUSSIAN, 10) == 4


def test_dense_draws_are_uniform(oracle, table):
    p = oracle.make_params(oracle.MODE_TABLE, 4000, 1, 77, table=table)
    cnt = np.zeros(table.size)
    for path in range(300):
        cnt += np.bincount(oracle.counter_path_indices(p, path), minlength=table.size)
    e = cnt.sum() / table.size
    chi = ((cnt - e) ** 2 / e).sum()
    assert abs(chi - (table.size - 1)) < 5 * np.sqrt(2 * (table.size - 1))
