"""The host-buffer pipeline (BASELINE configs[4]: P = 1000, final values to host memory, D2H on a side
stream; reference pattern src/simulations.cu:615-626) against the ORACLE, its pinning policies and
progress reporting, and the engine's stream discipline under torch."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED5EED5EED5EED


def _fresh_engine(table, monkeypatch, **env):
    import stock_market_monte_carlo_amd as S
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    e = S.Engine(0)  # the knobs are read when an engine is created
    e.set_table(table)
    return e


@pytest.mark.parametrize("mode_name", ["gaussian", "table"])
def test_config4_path_many_host_chunks_against_the_oracle(table, oracle, monkeypatch, mode_name):
    """P = 1000, several host chunks (chunk shrunk to 4096 paths so the oracle finishes in seconds),
    ragged tail, statistics and chunk means merged over chunks: bit-exact final values."""
    import stock_market_monte_carlo_amd as S
    mode = S.MODE_GAUSSIAN if mode_name == "gaussian" else S.MODE_TABLE
    eng = _fresh_engine(table, monkeypatch, SMMC_HOST_CHUNK_PATHS=4096)
    n, p, first = 3 * 4096 + 77, 1000, 10 ** 9 - 5000  # ids as the last rank of a 1e9-path run sees them
    sim = S.Engine.make_sim(n, p, mode, SEED, first_path=first, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
    pinned = __import__("torch").empty(n, dtype=__import__("torch").float32, pin_memory=True).numpy()
    prog = C.c_int64(-1)
    host, st, (cm, cv) = eng.simulate_to_host(sim, out=pinned, want_stats=True, want_chunk_stats=True, progress=prog)
    o = oracle.counter_mc(oracle.make_params(mode, p, n, SEED, first_path=first, table=table, n_bins=100, hist_lo=0.0,
                                             hist_hi=20000.0))
    assert np.array_equal(host.view(np.uint32), o["final"].view(np.uint32))
    assert prog.value == n
    assert st.count == n and st.below == o["stats"].below and np.array_equal(st.hist, o["hist"])
    assert st.sum == pytest.approx(o["stats"].sum, rel=1e-12) and st.sumsq == pytest.approx(o["stats"].sumsq, rel=1e-12)
    assert st.min == o["stats"].min and st.max == o["stats"].max
    # chunk means: 4096 is a multiple of 256, so the per-256 chunks line up with a single launch's
    ocm, ocv = oracle.chunk_mean_var(o["final"])
    np.testing.assert_allclose(cm, ocm, rtol=1e-6)
    np.testing.assert_allclose(cv, ocv, rtol=1e-5, atol=1e-30)
    # pageable destination, no statistics: the same bits
    host2, _, _ = eng.simulate_to_host(sim)
    assert np.array_equal(host2.view(np.uint32), o["final"].view(np.uint32))
    eng.close()


def test_config4_full_chunk_size_spot_check(table, oracle):
    """The real chunk length (2^22 paths) at P = 1000: eight chunks + a tail into pinned memory; the
    first, chunk-boundary and last paths against the oracle, the rest by count conservation."""
    import torch
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    n, p = (1 << 25) + 1234, 1000
    sim = S.Engine.make_sim(n, p, S.MODE_GAUSSIAN, SEED, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
    pinned = torch.empty(n, dtype=torch.float32, pin_memory=True).numpy()
    host, st, _ = eng.simulate_to_host(sim, out=pinned, want_stats=True)
    assert st.count == n and int(st.hist.sum()) + st.underflow + st.overflow == n
    for first in (0, (1 << 22) - 100, (1 << 24) - 100, (1 << 25) - 100, n - 200):
        o = oracle.counter_mc(oracle.make_params(oracle.MODE_GAUSSIAN, p, 200, SEED, first_path=first))
        assert np.array_equal(host[first:first + 200].view(np.uint32), o["final"].view(np.uint32)), first
    assert float(host.astype(np.float64).sum()) == pytest.approx(st.sum, rel=1e-12)
    eng.close()


@pytest.mark.parametrize("policy", ["whole", "chunk"])
def test_pinning_policies_do_not_change_results(table, oracle, monkeypatch, policy):
    """SMMC_PIN_HOST: the caller's pageable buffer is page-locked for the call (whole, or chunk by chunk
    ahead of the copies) and released again; results and the buffer's usability are unchanged."""
    import stock_market_monte_carlo_amd as S
    eng = _fresh_engine(table, monkeypatch, SMMC_PIN_HOST=policy, SMMC_HOST_CHUNK_PATHS=1 << 22)
    n, p = (1 << 23) + (1 << 22) + 999, 4  # 48 MiB + a tail: above the 32 MiB threshold, 3 chunks + tail
    sim = S.Engine.make_sim(n, p, S.MODE_TABLE, 11)
    out = np.full(n, -1.0, dtype=np.float32)
    host, _, _ = eng.simulate_to_host(sim, out=out)
    dev = eng.simulate(sim).final.cpu().numpy()
    assert np.array_equal(host.view(np.uint32), dev.view(np.uint32))
    for first in (0, (1 << 22) - 50, n - 100):
        o = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, p, 100, 11, first_path=first, table=table))
        assert np.array_equal(host[first:first + 100].view(np.uint32), o["final"].view(np.uint32))
    # the registration is gone: a second run (re-registers) and plain host use both work
    out[:] = 0.0
    host, _, _ = eng.simulate_to_host(sim, out=out)
    assert np.array_equal(host.view(np.uint32), dev.view(np.uint32))
    eng.close()


@pytest.mark.parametrize("n,periods", [(20_000_000, 8), (1_000_000, 360)])
def test_progress_advances_in_steps(table, n, periods):
    """With a progress pointer the chunks shrink to ~N/16 (>= 2^16 paths): a poller sees the counter move
    -- also at BASELINE configs[0] size (1e6 paths x 360 periods), where round 2's 2^20-path floor made
    the counter jump 0 -> N (the reference adds 1000 paths at a time, src/simulations.cpp:254)."""
    import threading
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    eng.set_table(table)
    sim = S.Engine.make_sim(n, periods, S.MODE_TABLE, 5)
    prog = C.c_int64(0)
    seen, stop = set(), threading.Event()

    def poll():
        while not stop.is_set():
            seen.add(prog.value)

    t = threading.Thread(target=poll)
    t.start()
    eng.simulate_to_host(sim, progress=prog)
    stop.set()
    t.join()
    assert prog.value == n
    mid = sorted(v for v in seen if 0 < v < n)
    assert len(mid) >= 3 and all(v % 1024 == 0 for v in mid)
    eng.close()


def test_polled_and_unpolled_runs_agree_except_in_the_last_bits_of_the_sums(table):
    """polling shortens the chunks, and the statistics record is merged over chunks.  What the
    header promises: final values, counters, min / max and histogram do not depend on it; sum and sumsq are
    double sums in chunk order -- equal to 1e-13, not necessarily bit for bit."""
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    eng.set_table(table)
    n = 40_000_000 + 321
    sim = S.Engine.make_sim(n, 8, S.MODE_TABLE, 9, n_bins=64, hist_lo=500.0, hist_hi=2000.0)
    plain, st_plain, _ = eng.simulate_to_host(sim, want_stats=True)
    prog = C.c_int64(0)
    polled, st_polled, _ = eng.simulate_to_host(sim, want_stats=True, progress=prog)
    assert prog.value == n and np.array_equal(plain.view(np.uint32), polled.view(np.uint32))
    assert (st_plain.count, st_plain.below, st_plain.underflow, st_plain.overflow, st_plain.min, st_plain.max) == \
           (st_polled.count, st_polled.below, st_polled.underflow, st_polled.overflow, st_polled.min, st_polled.max)
    assert np.array_equal(st_plain.hist, st_polled.hist)
    assert st_polled.sum == pytest.approx(st_plain.sum, rel=1e-13) and st_polled.sumsq == pytest.approx(st_plain.sumsq, rel=1e-13)
    eng.close()


def test_chunk_pinning_on_a_buffer_that_is_not_page_aligned(table, oracle, monkeypatch, capfd):
    """SMMC_PIN_HOST=chunk on a result buffer that starts in the middle of a page: neighbouring chunks share
    pages, every page has one owning chunk, and no registration falls back (SMMC_VERBOSE would say so)."""
    import stock_market_monte_carlo_amd as S
    eng = _fresh_engine(table, monkeypatch, SMMC_PIN_HOST="chunk", SMMC_HOST_CHUNK_PATHS=1 << 21, SMMC_VERBOSE=1)
    n, p = 5 * (1 << 21) + 777, 4  # 40 MiB + a tail: six chunks
    backing = np.full(n + 4096, -1.0, dtype=np.float32)
    off = (-(backing.ctypes.data // 4) % 1024 + 100) % 1024 + 1  # 4-byte aligned, never on a page boundary
    out = backing[off:off + n]
    assert out.ctypes.data % 4096 != 0
    sim = S.Engine.make_sim(n, p, S.MODE_TABLE, 11)
    host, _, _ = eng.simulate_to_host(sim, out=out)
    dev = eng.simulate(sim).final.cpu().numpy()
    assert np.array_equal(host.view(np.uint32), dev.view(np.uint32))
    assert backing[off - 1] == -1.0 and backing[off + n] == -1.0
    assert "hipHostRegister" not in capfd.readouterr().err
    host2, _, _ = eng.simulate_to_host(sim, out=out)  # every page was released: registers again
    assert np.array_equal(host2.view(np.uint32), dev.view(np.uint32))
    assert "hipHostRegister" not in capfd.readouterr().err
    eng.close()


def test_engine_follows_the_current_torch_stream(table, oracle):
    """an engine built while one stream was current must launch on the stream that is
    current at CALL time (outputs are allocated there), and results consumed on that stream are right."""
    import torch
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)  # built on the default stream
    eng.set_table(table)
    sim = S.Engine.make_sim(50_000, 36, S.MODE_TABLE, 77, n_bins=10, hist_lo=0.0, hist_hi=5000.0)
    want = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 36, 50_000, 77, table=table))["final"]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        r = eng.simulate(sim, want_stats=True)
        assert eng.stream_handle() == side.cuda_stream
        total = r.final.double().sum()  # consumed on `side`, no explicit synchronisation
        rec = eng.values_stats(r.final, 1000.0, 10, 0.0, 5000.0)
    side.synchronize()
    assert np.array_equal(r.final.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert float(total) == pytest.approx(float(want.astype(np.float64).sum()), rel=1e-12)
    assert eng.read_stats(rec).count == 50_000
    r2 = eng.simulate(sim)  # back on the default stream
    assert eng.stream_handle() == torch.cuda.current_stream().cuda_stream
    assert np.array_equal(r2.final.cpu().numpy().view(np.uint32), want.view(np.uint32))
    eng.close()


def test_engine_owned_stream_is_ordered_with_torch(table, oracle):
    import torch
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0, stream="new")
    eng.set_table(table)
    sim = S.Engine.make_sim(50_000, 36, S.MODE_TABLE, 77)
    want = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 36, 50_000, 77, table=table))["final"]
    for _ in range(3):
        r = eng.simulate(sim)
        got = r.final.clone()  # torch's stream waits for the engine's
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))
    eng.close()


def test_a_stream_destroyed_while_torch_still_holds_the_outputs(table, oracle):
    """Round 2's segfault: the engine's own stream was wrapped for torch's caching allocator,
    which then polled events on a stream the engine had destroyed.  Now ordering is by events at call time
    only -- destroy the engine (and its stream) while the output tensors are still alive and unread, then
    use, free and re-allocate them."""
    import gc
    import torch
    import stock_market_monte_carlo_amd as S
    want = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 36, 50_000, 77, table=table))["final"]
    sim = S.Engine.make_sim(50_000, 36, S.MODE_TABLE, 77, n_bins=10, hist_lo=0.0, hist_hi=5000.0)
    held = []
    for _ in range(3):
        eng = S.Engine(0, stream="new")
        eng.set_table(table)
        r = eng.simulate(sim, want_stats=True, want_chunk_stats=True)
        eng.close()  # drains and destroys the engine-owned stream; r's tensors are still held, not yet read
        held.append(r)
    for r in held:
        assert np.array_equal(r.final.cpu().numpy().view(np.uint32), want.view(np.uint32))
    del held, r
    gc.collect()
    torch.cuda.empty_cache()  # the allocator frees the blocks: nothing refers to the dead streams
    x = torch.empty(1 << 22, device="cuda").normal_()
    torch.cuda.synchronize()
    assert torch.isfinite(x).all()
    # a caller's stream that dies between two calls of a "torch" engine: the next call re-binds and runs
    eng = S.Engine(0)
    eng.set_table(table)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        r = eng.simulate(sim)
    side.synchronize()
    del side
    gc.collect()
    r2 = eng.simulate(sim)
    eng.sync()
    assert np.array_equal(r2.final.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert np.array_equal(r.final.cpu().numpy().view(np.uint32), want.view(np.uint32))
    eng.close()
