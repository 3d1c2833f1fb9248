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
    """The real chunk length (2^24 paths) at P = 1000: two chunks + a tail into pinned memory; the
    first, the chunk-boundary and the last paths against the oracle, the rest by count conservation."""
    import torch
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    n, p = (1 << 25) + 1234, 1000
    sim = S.Engine.make_sim(n, p, S.MODE_GAUSSIAN, SEED, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
    pinned = torch.empty(n, dtype=torch.float32, pin_memory=True).numpy()
    host, st, _ = eng.simulate_to_host(sim, out=pinned, want_stats=True)
    assert st.count == n and int(st.hist.sum()) + st.underflow + st.overflow == n
    for first in (0, (1 << 24) - 100, (1 << 25) - 100, n - 200):
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


def test_progress_advances_in_steps(table):
    """With a progress pointer the chunks shrink to ~N/16 (>= 2^20 paths): a poller sees the counter move."""
    import threading
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    eng.set_table(table)
    n = 20_000_000
    sim = S.Engine.make_sim(n, 8, S.MODE_TABLE, 5)
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
    assert len(mid) >= 3 and all(v % 256 == 0 for v in mid)
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
