"""The multi-device C entry (smmc_group_*, include/smmc.h; reference: mc_simulations_multi_gpu_launcher_async,
src/simulations.cu:576-655) on a one-GPU box: several shards on the one device with the host merge, and
the RCCL merge with the one rank a single device allows -- ncclCommInitAll, the grouped all-reduce of the
integer record and the device-resident merged record are the code that runs on 8 GPUs."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED5EED5EED5EED


def _sim(S, n, p, mode, **kw):
    return S.Engine.make_sim(n, p, mode, SEED, n_bins=100, hist_lo=0.0, hist_hi=20000.0, **kw)


@pytest.mark.parametrize("mode_name", ["gaussian", "table"])
def test_three_shards_on_one_device_equal_the_oracle(table, oracle, mode_name):
    import stock_market_monte_carlo_amd as S
    mode = S.MODE_GAUSSIAN if mode_name == "gaussian" else S.MODE_TABLE
    n, p = 100_003, 360  # not divisible by 3: the remainder is kept (src/simulations.cu:602-603 drops it)
    g = S.Group([0, 0, 0], merge="host")
    g.set_table(table)
    assert len(g) == 3 and [g.shard(n, i) for i in range(3)] == [(0, 33335), (33335, 33334), (66669, 33334)]
    prog = C.c_int64(-1)
    host, st, _ = g.simulate(_sim(S, n, p, mode), want_stats=True, progress=prog)
    o = oracle.counter_mc(oracle.make_params(mode, p, n, SEED, table=table, n_bins=100, hist_lo=0.0, hist_hi=20000.0))
    assert np.array_equal(host.view(np.uint32), o["final"].view(np.uint32))
    assert prog.value == n and st.count == n and st.below == o["stats"].below and np.array_equal(st.hist, o["hist"])
    assert (st.underflow, st.overflow, st.min, st.max) == (o["stats"].underflow, o["stats"].overflow, o["stats"].min, o["stats"].max)
    assert st.sum == pytest.approx(o["stats"].sum, rel=1e-12) and st.sumsq == pytest.approx(o["stats"].sumsq, rel=1e-12)
    # statistics only: no final values cross PCIe (BASELINE configs[3])
    _, st2, _ = g.simulate(_sim(S, n, p, mode), want_final=False, want_stats=True)
    assert np.array_equal(st2.hist, st.hist) and st2.below == st.below and st2.sum == st.sum
    # chunk means need shards that start on a multiple of 256 paths
    with pytest.raises(S.SmmcError, match="multiple of 256"):
        g.simulate(_sim(S, n, p, mode), want_chunk_stats=True)
    n2 = 3 * 256 * 50
    host2, _, (cm, cv) = g.simulate(_sim(S, n2, 36, mode), want_chunk_stats=True)
    o2 = oracle.counter_mc(oracle.make_params(mode, 36, n2, SEED, table=table))
    ocm, ocv = oracle.chunk_mean_var(o2["final"])
    assert np.array_equal(host2.view(np.uint32), o2["final"].view(np.uint32))
    np.testing.assert_allclose(cm, ocm, rtol=1e-6)
    np.testing.assert_allclose(cv, ocv, rtol=1e-5, atol=1e-30)
    g.close()


def test_rccl_merge_equals_the_host_merge_bit_for_bit(table):
    """G = 1 through RCCL on the one GPU: communicator created once and reused, the
    merged record identical to the host merge, and the merged integer record readable in HBM."""
    import stock_market_monte_carlo_amd as S
    host_g = S.Group([0], merge="host")
    rccl_g = S.Group([0], merge="rccl")
    for g in (host_g, rccl_g):
        g.set_table(table)
    e_ms, c_ms, _ = rccl_g.timings()
    assert c_ms > 0.0 and host_g.timings()[1] == 0.0
    merges = []
    for n, p, mode in ((200_000, 360, S.MODE_GAUSSIAN), (50_001, 36, S.MODE_TABLE), (0, 8, S.MODE_TABLE)):
        sim = _sim(S, n, p, mode)
        fa, sa, _ = host_g.simulate(sim, want_stats=True)
        fb, sb, _ = rccl_g.simulate(sim, want_stats=True)
        assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
        assert (sa.count, sa.below, sa.underflow, sa.overflow) == (sb.count, sb.below, sb.underflow, sb.overflow)
        assert np.array_equal(sa.hist, sb.hist)
        assert np.float64(sa.sum).tobytes() == np.float64(sb.sum).tobytes()
        assert np.float64(sa.sumsq).tobytes() == np.float64(sb.sumsq).tobytes()
        assert np.float32(sa.min).tobytes() == np.float32(sb.min).tobytes() and np.float32(sa.max).tobytes() == np.float32(sb.max).tobytes()
        merges.append(rccl_g.timings()[2])
        if n:
            # the merged record in the device's own memory: counters, then (after the 64-byte header) the buckets
            ptr = rccl_g.device_record(0)
            assert ptr != 0
            raw = np.empty(64 + 8 * 100, dtype=np.uint8)
            hip = C.CDLL("libamdhip64.so.7")  # the runtime already in the process (same soname)
            assert hip.hipMemcpy(C.c_void_p(raw.ctypes.data), C.c_void_p(ptr), C.c_size_t(raw.size), 2) == 0  # device to host
            assert raw[:32].view(np.uint64).tolist() == [sb.count, sb.below, sb.underflow, sb.overflow]
            assert np.array_equal(raw[64:].view(np.uint64), sb.hist)
    assert all(m >= 0.0 for m in merges)
    assert rccl_g.timings()[1] == c_ms  # the communicator was not created again
    with pytest.raises(S.SmmcError, match="SMMC_MERGE_RCCL"):
        host_g.device_record(0)
    host_g.close()
    rccl_g.close()


def test_group_argument_errors():
    import stock_market_monte_carlo_amd as S
    with pytest.raises(S.SmmcError, match="distinct devices"):
        S.Group([0, 0], merge="rccl")
    with pytest.raises(S.SmmcError, match="out of range"):
        S.Group([0, 99], merge="host")
    g = S.Group([0])
    with pytest.raises(S.SmmcError, match="set_table"):
        g.simulate(S.Engine.make_sim(10, 4, S.MODE_TABLE, 1))
    g.close()


def test_reference_stream_through_a_group(table, oracle):
    import stock_market_monte_carlo_amd as S
    g = S.Group([0, 0])
    g.set_table(table)
    n, p, seed0 = 50_001, 360, 77
    sim = S.Engine.make_sim(n, p, S.MODE_TABLE, seed0, stream="ref", n_bins=20, hist_lo=0.0, hist_hi=30000.0)
    host, st, _ = g.simulate(sim, want_stats=True)
    want, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0)
    assert np.array_equal(host.view(np.uint32), want.view(np.uint32))
    ost, ohist = oracle.values_stats(want, 1000.0, 20, 0.0, 30000.0)
    assert np.array_equal(st.hist, ohist) and st.below == ost.below
    g.close()
