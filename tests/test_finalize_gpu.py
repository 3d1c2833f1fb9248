"""The statistics header is written by the LAST workgroup of paths_kernel (round 4; round 3 launched
finalize_kernel behind it): same record, bit for bit, as the separate launch gives (SMMC_FINALIZE=launch), the
finished-workgroup counter is left at zero for the next launch, and capping a small launch at the resident
workgroups (SMMC_SMALL_LAUNCH_ROUNDS) changes no per-path value and no integer of the record."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED5EED5EED5EED


def _engine(table, monkeypatch, **env):
    import stock_market_monte_carlo_amd as S
    for k in ("SMMC_FINALIZE", "SMMC_SMALL_LAUNCH_ROUNDS"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    e = S.Engine(0)  # the knobs are read when an engine is created
    e.set_table(table)
    return e


def _record(eng, sim, **kw):
    r = eng.simulate(sim, want_stats=True, **kw)
    eng.sync()
    return r.stats_raw.cpu().numpy().tobytes(), r


@pytest.mark.parametrize("mode_name", ["gaussian", "table"])
def test_last_workgroup_fold_equals_the_finalize_launch_bit_for_bit(table, oracle, monkeypatch, mode_name):
    import stock_market_monte_carlo_amd as S
    mode = S.MODE_GAUSSIAN if mode_name == "gaussian" else S.MODE_TABLE
    fused = _engine(table, monkeypatch, SMMC_SMALL_LAUNCH_ROUNDS=0)
    apart = _engine(table, monkeypatch, SMMC_SMALL_LAUNCH_ROUNDS=0, SMMC_FINALIZE="launch")
    try:
        # one workgroup, a partial chunk, fewer workgroups than fold accumulators, more, the full grid (64 per CU)
        for n, p in ((1, 7), (255, 360), (256, 360), (4099, 360), (300_001, 360), (1_000_000, 360), (6_000_000, 36), (50_000_000, 8)):
            sim = S.Engine.make_sim(n, p, mode, SEED, first_path=3, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
            a, ra = _record(fused, sim, want_final=True, want_chunk_stats=True)
            b, rb = _record(apart, sim, want_final=True, want_chunk_stats=True)
            assert a == b, (n, p)
            assert np.array_equal(ra.final.cpu().numpy().view(np.uint32), rb.final.cpu().numpy().view(np.uint32))
            st = S.engine.stats_from_bytes(a)
            assert st.count == n and int(st.hist.sum()) + st.underflow + st.overflow == n
            if n <= 300_001:
                o = oracle.counter_mc(oracle.make_params(mode, p, n, SEED, first_path=3, table=table, n_bins=100, hist_lo=0.0,
                                                         hist_hi=20000.0))
                assert st.below == o["stats"].below and np.array_equal(st.hist, o["hist"])
                assert st.min == o["stats"].min and st.max == o["stats"].max
                assert st.sum == pytest.approx(o["stats"].sum, rel=1e-12) and st.sumsq == pytest.approx(o["stats"].sumsq, rel=1e-12)
        # statistics only (no final values, no chunk outputs): the same record
        sim = S.Engine.make_sim(1_000_000, 360, mode, SEED, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
        assert _record(fused, sim, want_final=False)[0] == _record(apart, sim, want_final=False)[0]
        # an empty run still has a record (count 0, min +inf, max -inf)
        empty = S.engine.stats_from_bytes(_record(fused, S.Engine.make_sim(0, 360, mode, SEED, n_bins=4, hist_lo=0.0, hist_hi=1.0))[0])
        assert empty.count == 0 and empty.min == np.inf and empty.max == -np.inf and int(empty.hist.sum()) == 0
    finally:
        fused.close()
        apart.close()


def test_the_counter_is_ready_for_the_next_launch(table, monkeypatch):
    """Sixty launches of different sizes enqueued back to back on one engine (no host synchronisation between
    them), the same launch interleaved: every repetition gives the same record -- the folding workgroup left the
    counter at zero and no launch folded before all its workgroups had finished."""
    import stock_market_monte_carlo_amd as S
    eng = _engine(table, monkeypatch)
    try:
        ref_sim = S.Engine.make_sim(777_777, 360, S.MODE_GAUSSIAN, SEED, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
        want, _ = _record(eng, ref_sim, want_final=False)
        pending = []
        for i in range(60):
            other = S.Engine.make_sim(1000 + 37_003 * (i % 7), 36 + i, S.MODE_TABLE if i % 2 else S.MODE_GAUSSIAN, i, n_bins=10,
                                      hist_lo=0.0, hist_hi=5000.0)
            eng.simulate(other, want_final=False, want_stats=True)
            pending.append(eng.simulate(ref_sim, want_final=False, want_stats=True).stats_raw)
        eng.sync()
        assert all(r.cpu().numpy().tobytes() == want for r in pending)
    finally:
        eng.close()


def test_two_engines_on_their_own_streams_fold_independently(table, monkeypatch):
    import threading
    import stock_market_monte_carlo_amd as S
    for k in ("SMMC_FINALIZE", "SMMC_SMALL_LAUNCH_ROUNDS"):
        monkeypatch.delenv(k, raising=False)
    engines = [S.Engine(0, stream="new") for _ in range(3)]
    sims = [S.Engine.make_sim(2_000_000 + 1000 * i, 360, S.MODE_GAUSSIAN, SEED + i, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
            for i in range(3)]
    try:
        want = [_record(e, s, want_final=False)[0] for e, s in zip(engines, sims)]
        got = [[] for _ in engines]

        def run(i):
            for _ in range(20):
                got[i].append(engines[i].simulate(sims[i], want_final=False, want_stats=True).stats_raw)
            engines[i].sync()

        threads = [threading.Thread(target=run, args=(i,)) for i in range(3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for i in range(3):
            assert all(r.cpu().numpy().tobytes() == want[i] for r in got[i]), i
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("mode_name", ["gaussian", "table"])
def test_capping_a_small_launch_changes_no_value(table, monkeypatch, mode_name):
    """BASELINE configs[0]'s size (1e6 paths: 3907 chunks) and a few around the cap's window: final values, chunk
    means / variances, counters, buckets, min / max identical with the cap (default) and without; the double sums
    are sums over another grid's partials and agree to 1e-12."""
    import stock_market_monte_carlo_amd as S
    mode = S.MODE_GAUSSIAN if mode_name == "gaussian" else S.MODE_TABLE
    capped = _engine(table, monkeypatch)
    plain = _engine(table, monkeypatch, SMMC_SMALL_LAUNCH_ROUNDS=0)
    try:
        for n in (200_000, 262_144 + 5, 1_000_000, 2_500_000, 9_000_000):
            sim = S.Engine.make_sim(n, 360 if n <= 2_500_000 else 36, mode, SEED, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
            a, ra = _record(capped, sim, want_final=True, want_chunk_stats=True)
            b, rb = _record(plain, sim, want_final=True, want_chunk_stats=True)
            assert np.array_equal(ra.final.cpu().numpy().view(np.uint32), rb.final.cpu().numpy().view(np.uint32)), n
            assert np.array_equal(ra.chunk_mean.cpu().numpy().view(np.uint32), rb.chunk_mean.cpu().numpy().view(np.uint32))
            assert np.array_equal(ra.chunk_var.cpu().numpy().view(np.uint32), rb.chunk_var.cpu().numpy().view(np.uint32))
            sa, sb = S.engine.stats_from_bytes(a), S.engine.stats_from_bytes(b)
            assert (sa.count, sa.below, sa.underflow, sa.overflow, sa.min, sa.max) == (sb.count, sb.below, sb.underflow, sb.overflow, sb.min, sb.max)
            assert np.array_equal(sa.hist, sb.hist)
            assert sa.sum == pytest.approx(sb.sum, rel=1e-12) and sa.sumsq == pytest.approx(sb.sumsq, rel=1e-12)
    finally:
        capped.close()
        plain.close()
