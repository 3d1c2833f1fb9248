"""Pins the CPU oracle (oracle/smmc_oracle.c) to everything that can pin it here.

The reference has no tests or fixtures and could not be built (oracle/Makefile), so
"parity unpinned" holds against an executed reference.  These tests pin:
  * mt19937 / uniform_int_distribution<int>  -> system libstdc++ outputs
    (tests/golden/libstdcxx_random.json, made by oracle/pin/pin_libstdcxx.cpp) and the
    ISO C++ known answer;
  * whole reference-style paths (src/simulations.cpp:240-252 with explicit seeds)
    -> the same fixture;
  * Philox4x32-10 -> published Random123 known-answer vectors;
  * update_fund -> hand-computable IEEE cases.
"""
import json
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _pin():
    with open(os.path.join(HERE, "golden", "libstdcxx_random.json")) as f:
        return json.load(f)


def test_mt19937_iso_known_answer(oracle):
    # ISO C++ [rand.predef]: 10000th consecutive invocation of a default mt19937
    assert int(oracle.mt19937_raw(5489, 10000)[-1]) == 4123659995
    assert _pin()["mt19937_default_10000th"] == 4123659995


def test_mt19937_raw_matches_libstdcxx(oracle):
    for case in _pin()["mt19937_raw"]:
        got = oracle.mt19937_raw(case["seed"], len(case["out"]))
        assert [int(x) for x in got] == case["out"], case["seed"]


def test_lemire_index_matches_libstdcxx(oracle):
    # includes range 1 (always 0), powers of two, and 2^31-1 where rejection is common
    for case in _pin()["uniform_int"]:
        got = oracle.mt19937_indices(case["seed"], case["range"], len(case["out"]))
        assert [int(x) for x in got] == case["out"], (case["seed"], case["range"])


def test_reference_style_paths_match_libstdcxx(oracle, table):
    pin = _pin()
    assert pin["table_len"] == table.size
    for case in pin["paths"]:
        # engine (R) seeds path id with seed0 + id; run 32 paths
        got, _ = oracle.ref_mc_simulations(32, case["n_periods"], case["initial_capital"], table,
                                           case["seed0"], n_threads=1)
        assert [int(x) for x in got.view(np.uint32)] == case["final_bits"], case["n_periods"]


def test_paths_in_which_the_real_library_rejects_match(oracle, table):
    """The fixture's `rejecting_paths`: seeds whose path made the system libstdc++'s uniform_int_distribution
    reject a generator output (one engine call more than periods).  The oracle's hand-written Lemire map must
    continue exactly as the library does after a rejection."""
    cases = _pin()["rejecting_paths"]
    assert len(cases) == 12
    for c in cases:
        got, _ = oracle.ref_mc_simulations(1, c["n_periods"], c["initial_capital"], table, c["seed"], n_threads=1)
        assert int(got.view(np.uint32)[0]) == c["final_bits"], c
        # and the oracle really rejects there: its index stream consumes one output more than it yields
        raw = oracle.mt19937_raw(c["seed"], c["engine_calls"])
        low = (raw.astype(np.uint64) * table.size) & 0xFFFFFFFF
        assert int((low < (2 ** 32 - table.size) % table.size).sum()) == c["engine_calls"] - c["n_periods"]


def test_trajectories_match_libstdcxx(oracle, table):
    """`trajectories` of the fixture (mc_simulations_keepdata's sample_returns_historical + many_updates, computed by the
    system libstdc++): the oracle's index stream + many_updates give every value."""
    for c in _pin()["trajectories"]:
        idx = oracle.mt19937_indices(c["seed"], table.size, c["n_periods"])
        row = oracle.many_updates(c["initial_capital"], table[idx], c["n_periods"])
        assert [int(x) for x in row.view(np.uint32)] == c["value_bits"], (c["n_periods"], c["seed"])


def test_reference_engine_thread_count_invariant(oracle, table):
    a, _ = oracle.ref_mc_simulations(5000, 36, 1000.0, table, 77, n_threads=1)
    b, used = oracle.ref_mc_simulations(5000, 36, 1000.0, table, 77, n_threads=4)
    assert used == 4
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0],
         [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kat:
        assert [int(x) for x in oracle.philox4x32_10(ctr, key)] == want


def _bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def test_update_fund_ieee_cases(oracle):
    # src/simulations.cpp:14-16: fund * (100.0f + r) / 100
    assert _bits(oracle.update_fund(1000.0, 0.5)) == 0x447B4000  # 1005.0
    assert oracle.update_fund(1000.0, 0.0) == 1000.0
    assert oracle.update_fund(1000.0, -100.0) == 0.0
    assert oracle.update_fund(0.0, 5.0) == 0.0
    assert oracle.update_fund(1000.0, 100.0) == 2000.0
    # against numpy float32 arithmetic (three separate roundings)
    rng = np.random.default_rng(1)
    f = rng.uniform(1, 1e6, 2000).astype(np.float32)
    r = rng.normal(0.6, 4.3, 2000).astype(np.float32)
    want = (f * (np.float32(100.0) + r)) / np.float32(100.0)
    got = np.array([oracle.update_fund(float(a), float(b)) for a, b in zip(f, r)], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_many_updates_shape_and_chain(oracle):
    # src/simulations.cpp:18-39: n_periods + 1 outputs, totals[0] = start value
    rets = np.array([1.0, -2.0, 3.5, 0.0], dtype=np.float32)
    out = oracle.many_updates(1000.0, rets, 4)
    assert out.shape == (5,) and out[0] == 1000.0
    t = np.float32(1000.0)
    for i, r in enumerate(rets):
        t = (t * (np.float32(100.0) + r)) / np.float32(100.0)
        assert _bits(float(out[i + 1])) == _bits(float(t))
    assert oracle.many_updates(5.0, rets, 0).tolist() == [5.0]


def test_engine_R_equals_the_reference_loop_written_with_the_real_libstdcxx_classes(oracle, table):
    """oracle/asref_cpu.cpp is the reference's per-path loop (src/simulations.cpp:240-252) with
    std::mt19937 + std::uniform_int_distribution<int> themselves; seeded with seed0 + id instead of
    std::random_device it must give engine (R)'s bits -- the hand-written mt19937 and Lemire map checked
    against the library on whole paths, ragged block counts and thread counts."""
    for n, p, seed0, threads in ((2501, 37, 777, 2), (1000, 360, 0xFFFFFFF0, 3), (17, 1000, 5489, 1), (5, 0, 1, 1)):
        a, _ = oracle.asref_mc_simulations(n, p, 1000.0, table, n_threads=threads, fixed_seed0=seed0)
        b, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, seed0, n_threads=threads)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (n, p, seed0)
    # unseeded (the reference's behaviour): not reproducible, but the same distribution
    a, used = oracle.asref_mc_simulations(20000, 120, 1000.0, table, n_threads=2)
    b, _ = oracle.asref_mc_simulations(20000, 120, 1000.0, table, n_threads=2)
    assert used == 2 and not np.array_equal(a, b)
    r, _ = oracle.ref_mc_simulations(20000, 120, 1000.0, table, 1, n_threads=2)
    la, lr = np.log(a.astype(np.float64) / 1000.0), np.log(r.astype(np.float64) / 1000.0)
    se = lr.std() / np.sqrt(lr.size) * np.sqrt(2.0)
    assert abs(la.mean() - lr.mean()) < 5 * se and abs(la.std() / lr.std() - 1.0) < 0.05


import pytest


@pytest.mark.parametrize("stream", [2, 3])
def test_oracle_reproduces_the_frozen_counter_stream_vectors(oracle, table, stream):
    """tests/golden/counter_stream_v{2,3}.json were frozen from the oracle when the stream contracts were fixed (round 2 /
    round 3); the GPU suite compares the kernels with them, and the kernels with the oracle as it is built now.  This
    is the third side, on the CPU: the oracle AS BUILT NOW against the frozen vectors -- so that a change that moves the
    oracle and the kernels together (they share the generated Box-Muller tables, tools/gen_bm_tables.py) cannot pass
    unnoticed where there is no GPU."""
    with open(os.path.join(HERE, "golden", f"counter_stream_v{stream}.json")) as f:
        gold = json.load(f)
    modes = {"table": oracle.MODE_TABLE, "gaussian": oracle.MODE_GAUSSIAN}
    assert gold["stream"] == stream and len(gold["cases"]) == 24
    for c in gold["cases"]:
        p = oracle.make_params(modes[c["mode"]], c["n_periods"], c["n_paths"], seed=c["seed"], first_path=c["first_path"],
                               table=table, n_bins=c["n_bins"], hist_lo=c["hist_lo"], hist_hi=c["hist_hi"], stream=stream)
        r = oracle.counter_mc(p)
        st = r["stats"]
        assert [int(x) for x in r["final"].view(np.uint32)] == c["final_bits"], (c["mode"], c["n_periods"], c["first_path"])
        assert [int(x) for x in r["hist"]] == c["hist"]
        assert (int(st.below), int(st.underflow), int(st.overflow)) == (c["below"], c["underflow"], c["overflow"])
        assert st.sum == c["sum"] and st.sumsq == c["sumsq"] and float(st.min) == c["min"] and float(st.max) == c["max"]
