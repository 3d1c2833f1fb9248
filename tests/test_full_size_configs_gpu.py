"""BASELINE configs[3] and configs[4] at their REAL sizes, on one GPU.

configs[3]: Gaussian, 360 periods x 1e9 paths sharded over 8 GPUs, statistics record reduced.
configs[4]: Gaussian, 1000 periods x 1e9 paths over 8 GPUs, final values to pinned host memory, D2H
overlapped on a side stream (reference launcher: src/simulations.cu:599-626).

Eight MI355X are not available to the tests, but nothing in either configuration depends on WHICH device
runs a shard: a path is a function of (seed, global path id, parameters).  So one GPU runs every rank's
shard at the rank's real size with the rank's real ids -- one after another, and as eight concurrent shards
of one smmc_group -- and the results are compared with the whole 1e9-path run, with the oracle at every
shard boundary, and by the size-independent properties (count conservation, merged record == whole record)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED5EED5EED5EED
N_TOTAL, WORLD = 1_000_000_000, 8
BINS, LO, HI = 100, 0.0, 20000.0


def _oracle_final(oracle, periods, first, count):
    return oracle.counter_mc(oracle.make_params(oracle.MODE_GAUSSIAN, periods, count, SEED, first_path=first))["final"]


def test_config3_eight_shards_at_full_size_merge_to_the_whole_run(oracle):
    """1e9 x 360, statistics only, in ONE launch; then the eight dist.shard_range shares (1.25e8 paths each,
    ids from rank * 1.25e8) one after another WITH their final values: the first and last 100 paths of every
    share against the oracle, every share's record against a torch reduction of its own final values, and the
    eight records merged in rank order == the whole run's record (integers, buckets, min / max exactly; the
    double sums to 1e-12)."""
    import torch
    import stock_market_monte_carlo_amd as S
    from stock_market_monte_carlo_amd.dist import shard_range
    from stock_market_monte_carlo_amd.engine import merge_stats_bytes, stats_from_bytes
    eng = S.Engine(0)
    try:
        def make(first, count):
            return S.Engine.make_sim(count, 360, S.MODE_GAUSSIAN, SEED, first_path=first, n_bins=BINS, hist_lo=LO, hist_hi=HI)

        whole = eng.read_stats(eng.simulate(make(0, N_TOTAL), want_final=False, want_stats=True).stats_raw)
        assert whole.count == N_TOTAL and int(whole.hist.sum()) + whole.underflow + whole.overflow == N_TOTAL
        records, covered = [], 0
        final = torch.empty(N_TOTAL // WORLD + 1, dtype=torch.float32, device=eng.tdevice)
        for rank in range(WORLD):
            first, count = shard_range(N_TOTAL, WORLD, rank)
            assert first == covered
            covered += count
            r = eng.simulate(make(first, count), want_final=True, want_stats=True, out=final)
            eng.sync()
            records.append(r.stats_raw.cpu().numpy().tobytes())
            st = stats_from_bytes(records[-1])
            vals = final[:count]
            # both ends of the share: the ids a rank's neighbours end / begin with
            head, tail = vals[:100].cpu().numpy(), vals[count - 100:].cpu().numpy()
            assert np.array_equal(head.view(np.uint32), _oracle_final(oracle, 360, first, 100).view(np.uint32)), rank
            assert np.array_equal(tail.view(np.uint32), _oracle_final(oracle, 360, first + count - 100, 100).view(np.uint32)), rank
            # the share's record is the record of the values it stored
            assert st.count == count and st.below == int((vals < 1000.0).sum().item())
            assert st.sum == pytest.approx(float(vals.double().sum().item()), rel=1e-12)
            assert st.min == float(vals.min().item()) and st.max == float(vals.max().item())
            assert int(st.hist.sum()) + st.underflow + st.overflow == count
        assert covered == N_TOTAL
        merged = stats_from_bytes(merge_stats_bytes(records))
        assert merged.count == whole.count and merged.below == whole.below
        assert merged.underflow == whole.underflow and merged.overflow == whole.overflow
        assert np.array_equal(merged.hist, whole.hist)
        assert merged.min == whole.min and merged.max == whole.max
        assert merged.sum == pytest.approx(whole.sum, rel=1e-12) and merged.sumsq == pytest.approx(whole.sumsq, rel=1e-12)
        # 1e9 draws of the law: mean 1000 * 1.005^360 to 5 standard errors (relative sd 0.159)
        assert whole.mean == pytest.approx(1000.0 * 1.005 ** 360, rel=5 * 0.159 / np.sqrt(N_TOTAL) + 2e-6)
    finally:
        eng.close()


def test_config3_through_the_c_group_entry_eight_concurrent_shards():
    """benchmark_mc_gpu 8 360 1000000000's engine call at full size: ONE smmc_group of eight shards (eight
    engines, eight streams, eight host threads -- all on this box's one device), statistics only, host merge:
    the merged record equals the one-launch record."""
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    grp = S.Group([0] * WORLD)
    try:
        sim = S.Engine.make_sim(N_TOTAL, 360, S.MODE_GAUSSIAN, SEED, n_bins=BINS, hist_lo=LO, hist_hi=HI)
        whole = eng.read_stats(eng.simulate(sim, want_final=False, want_stats=True).stats_raw)
        _, st, _ = grp.simulate(sim, want_final=False, want_stats=True)
        assert [grp.shard(N_TOTAL, i) for i in range(WORLD)] == [(i * 125_000_000, 125_000_000) for i in range(WORLD)]
        assert st.count == N_TOTAL == whole.count and st.below == whole.below and np.array_equal(st.hist, whole.hist)
        assert st.underflow == whole.underflow and st.overflow == whole.overflow and st.min == whole.min and st.max == whole.max
        assert st.sum == pytest.approx(whole.sum, rel=1e-12) and st.sumsq == pytest.approx(whole.sumsq, rel=1e-12)
    finally:
        grp.close()
        eng.close()


def test_config4_at_full_size_whole_run_rank7_share_and_eight_shard_group(oracle):
    """1000 periods x 1e9 paths into 4 GB of pinned host memory through the chunked side-stream pipeline (239
    chunks of 2^22 paths); then rank 7's share alone -- 1.25e8 paths from id 8.75e8 into its own 500 MB pinned
    buffer, as that rank of an 8-GPU run would -- equal to the same ids of the whole run, bit for bit; then the
    eight shards as one smmc_group into a pageable 4 GB buffer (registered once by the group): equal again.
    Oracle: the first and last 200 paths of the run, both sides of every shard boundary, and both sides of two
    chunk boundaries inside rank 7's share."""
    import torch
    import stock_market_monte_carlo_amd as S
    from stock_market_monte_carlo_amd.dist import shard_range
    p = 1000
    eng = S.Engine(0)
    grp = None
    try:
        sim = S.Engine.make_sim(N_TOTAL, p, S.MODE_GAUSSIAN, SEED, n_bins=BINS, hist_lo=LO, hist_hi=HI)
        whole = torch.empty(N_TOTAL, dtype=torch.float32, pin_memory=True).numpy()
        whole[::1024] = -1.0  # touched, and recognisably not results
        _, st, _ = eng.simulate_to_host(sim, out=whole, want_stats=True)
        assert st.count == N_TOTAL and int(st.hist.sum()) + st.underflow + st.overflow == N_TOTAL
        assert float(whole.min()) == st.min > 0.0 and float(whole.max()) == st.max

        first7, count7 = shard_range(N_TOTAL, WORLD, 7)
        assert (first7, count7) == (875_000_000, 125_000_000)
        share = torch.empty(count7, dtype=torch.float32, pin_memory=True).numpy()
        sim7 = S.Engine.make_sim(count7, p, S.MODE_GAUSSIAN, SEED, first_path=first7, n_bins=BINS, hist_lo=LO, hist_hi=HI)
        _, st7, _ = eng.simulate_to_host(sim7, out=share, want_stats=True)
        assert np.array_equal(share.view(np.uint32), whole[first7:].view(np.uint32))
        assert st7.count == count7 and st7.below == int((share < np.float32(1000.0)).sum())

        chunk = 1 << 22
        spots = [0, N_TOTAL - 200]
        spots += [shard_range(N_TOTAL, WORLD, r)[0] - 100 for r in range(1, WORLD)]   # 100 paths either side of a shard boundary
        spots += [first7 + chunk - 100, first7 + 17 * chunk - 100]                    # ... and of two of rank 7's chunk boundaries
        spots += [7 * chunk - 100, 200 * chunk - 100]                                 # ... and of two of the whole run's
        for s in spots:
            want = _oracle_final(oracle, p, s, 200)
            assert np.array_equal(whole[s:s + 200].view(np.uint32), want.view(np.uint32)), s
            if s >= first7:
                assert np.array_equal(share[s - first7:s - first7 + 200].view(np.uint32), want.view(np.uint32)), s

        del share
        grp = S.Group([0] * WORLD)
        out = np.empty(N_TOTAL, dtype=np.float32)
        _, stg, _ = grp.simulate(sim, out=out, want_stats=True)
        assert np.array_equal(out.view(np.uint32), whole.view(np.uint32))
        assert stg.count == N_TOTAL and stg.below == st.below and np.array_equal(stg.hist, st.hist)
        assert stg.min == st.min and stg.max == st.max and stg.sum == pytest.approx(st.sum, rel=1e-12)
    finally:
        if grp is not None:
            grp.close()
        eng.close()
