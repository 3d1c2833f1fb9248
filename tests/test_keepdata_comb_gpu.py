"""keepdata, comb form (keepdata_comb_kernel): bit-exact trajectories against the oracle.

The comb kernel takes whole 2048-row super-chunks (64 streams of K consecutive rows, 32 rows apart,
per wave), stores every 128-byte line whole and once -- each stream from its first whole line through
the line its last row ends in, completed with the HEAD of the following row -- and leaves the last
< 2048 rows to the tile kernel.  Everything that can go wrong there is positional: which wave writes
which line, the extension into the next row, the first and the last line of the call, the hand-over to
the tile kernel, any base alignment.  So: every rows-per-stream setting, several phases, both draw
modes and all three draw schedules, guard bands around the buffer."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x5EED0123456789AB


@pytest.fixture(scope="module")
def eng(table):
    import stock_market_monte_carlo_amd as S
    e = S.Engine(0)
    e.set_table(table)
    yield e
    e.close()


def _run(eng, sim, n, p, shift, want_final=True):
    """keepdata through the C ABI into a guarded buffer at float offset `shift`."""
    import torch
    from stock_market_monte_carlo_amd import _lib
    buf = torch.full((n * (p + 1) + 96,), -7.0, dtype=torch.float32, device="cuda")
    view = buf[32 + shift:32 + shift + n * (p + 1)]
    fin = torch.full((n + 8,), -7.0, dtype=torch.float32, device="cuda") if want_final else None
    _lib.check(eng._L.smmc_engine_simulate_keepdata(eng._h, C.byref(sim), C.c_void_p(view.data_ptr()),
                                                    C.c_void_p(fin.data_ptr()) if want_final else None))
    eng.sync()
    got = buf.cpu().numpy()
    lo, hi = 32 + shift, 32 + shift + n * (p + 1)
    assert np.all(got[:lo] == -7.0) and np.all(got[hi:] == -7.0), "wrote outside the trajectory array"
    f = None
    if want_final:
        f = fin.cpu().numpy()
        assert np.all(f[n:] == -7.0)
        f = f[:n]
    return got[lo:hi].reshape(n, p + 1), f


@pytest.mark.parametrize("mode_name", ["table", "gaussian"])
@pytest.mark.parametrize("k_rows", [1, 2, 4, 8, 16, 32])
def test_comb_bit_exact_every_rows_per_stream(eng, oracle, table, monkeypatch, mode_name, k_rows):
    import stock_market_monte_carlo_amd as S
    mode = S.MODE_GAUSSIAN if mode_name == "gaussian" else S.MODE_TABLE
    monkeypatch.setenv("SMMC_KEEPDATA_KERNEL", "comb")
    monkeypatch.setenv("SMMC_KEEPDATA_K", str(k_rows))
    # two super-chunks + a tail for the tile kernel; phases 0 (whole first line), odd, 31
    for n, p, shift in ((2 * 2048 + 77, 72, 0), (2048, 64, 5), (2 * 2048 + 1, 360, 31), (2048 + 2047, 200, 17)):
        sim = S.Engine.make_sim(n, p, mode, SEED, first_path=11)
        traj, fin = _run(eng, sim, n, p, shift)
        o = oracle.counter_mc(oracle.make_params(mode, p, n, SEED, first_path=11, table=table), want_traj=True)
        bad = np.argwhere(traj.view(np.uint32) != o["traj"].view(np.uint32))
        assert bad.size == 0, (k_rows, n, p, shift, bad[:5].tolist())
        assert np.array_equal(fin.view(np.uint32), o["final"].view(np.uint32))


def test_comb_exact_multiple_of_a_super_chunk_has_no_row_to_extend_into(eng, oracle, table, monkeypatch):
    """n = 2048 m: the call's last row ends mid-line and nothing follows it; the guard band after the
    array must stay untouched (checked in _run) and the partial last line must hold the right values."""
    import stock_market_monte_carlo_amd as S
    monkeypatch.setenv("SMMC_KEEPDATA_KERNEL", "comb")
    for k_rows in (1, 4, 32):
        monkeypatch.setenv("SMMC_KEEPDATA_K", str(k_rows))
        for shift in (0, 9, 23):
            n, p = 4096, 64
            sim = S.Engine.make_sim(n, p, S.MODE_TABLE, 5)
            traj, fin = _run(eng, sim, n, p, shift)
            o = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, p, n, 5, table=table), want_traj=True)
            assert np.array_equal(traj.view(np.uint32), o["traj"].view(np.uint32)), (k_rows, shift)
            assert np.array_equal(fin.view(np.uint32), o["final"].view(np.uint32))


def test_comb_sparse_table_exact_divide_and_big_path_ids(eng, oracle, monkeypatch):
    """The four-draws-per-block table schedule (T > 2048), the IEEE-divide variant (a table the
    range proof rejects) and path ids beyond 2^32 that carry inside a stream."""
    import stock_market_monte_carlo_amd as S
    monkeypatch.setenv("SMMC_KEEPDATA_KERNEL", "comb")
    rng = np.random.default_rng(1)
    big = rng.normal(0.6, 4.3, 5000).astype(np.float32)
    wild = np.concatenate([rng.normal(0.6, 4.3, 1000), [250.0, -60.0]]).astype(np.float32)  # cannot be proven safe
    e2 = S.Engine(0)
    try:
        for tab, p in ((big, 64), (wild, 128)):
            e2.set_table(tab)
            n, first = 2048 + 300, (1 << 32) - 1000  # the 2^32 carry falls inside streams
            sim = S.Engine.make_sim(n, p, S.MODE_TABLE, SEED, first_path=first)
            traj, fin = _run(e2, sim, n, p, 3)
            o = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, p, n, SEED, first_path=first, table=tab), want_traj=True)
            assert np.array_equal(traj.view(np.uint32), o["traj"].view(np.uint32))
            assert np.array_equal(fin.view(np.uint32), o["final"].view(np.uint32))
    finally:
        e2.close()


def test_comb_is_chosen_for_large_calls_and_equals_the_tile_kernel(eng, table, monkeypatch):
    """Default selection at a size that fills the chip: same bits as the tile kernel over everything,
    final values equal the last column and the paths kernel's."""
    import torch
    import stock_market_monte_carlo_amd as S
    n, p = 2_000_000 + 1234, 360  # above the selection threshold (8 chunks of 64 rows per wave)
    for mode in (S.MODE_TABLE, S.MODE_GAUSSIAN):
        sim = S.Engine.make_sim(n, p, mode, SEED, first_path=7)
        monkeypatch.delenv("SMMC_KEEPDATA_KERNEL", raising=False)
        traj, fin = eng.simulate_keepdata(sim)
        monkeypatch.setenv("SMMC_KEEPDATA_KERNEL", "tile")
        traj_t, fin_t = eng.simulate_keepdata(sim)
        assert torch.equal(traj.view(torch.int32), traj_t.view(torch.int32))
        assert torch.equal(fin.view(torch.int32), fin_t.view(torch.int32))
        assert torch.equal(fin.view(torch.int32), traj[:, p].contiguous().view(torch.int32))
        assert torch.equal(fin.view(torch.int32), eng.simulate(sim).final.view(torch.int32))
        del traj, traj_t


def test_comb_is_not_used_where_it_does_not_apply(eng, oracle, table, monkeypatch):
    """Forcing the comb form on shapes it does not cover (n_periods not a multiple of the draws per
    block, short rows, fewer than 2048 rows) silently keeps the tile kernel: results stay right."""
    import stock_market_monte_carlo_amd as S
    monkeypatch.setenv("SMMC_KEEPDATA_KERNEL", "comb")
    for n, p in ((5000, 70), (5000, 361), (3000, 40), (2047, 64)):
        sim = S.Engine.make_sim(n, p, S.MODE_TABLE, SEED)
        traj, fin = _run(eng, sim, n, p, 1)
        o = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, p, n, SEED, table=table), want_traj=True)
        assert np.array_equal(traj.view(np.uint32), o["traj"].view(np.uint32)), (n, p)
        assert np.array_equal(fin.view(np.uint32), o["final"].view(np.uint32))
