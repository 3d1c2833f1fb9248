"""A plain C99 program (tests/c/c_caller.c) drives the C ABI the way INTEGRATION.md shows a binding: one engine
for statistics only, then smmc_group_* with several shards streaming final values into the caller's array and
ONE merged record.  Compiled with gcc -std=c99 -Wall -Wextra against include/smmc.h and libsmmc_hip.so --
no C++, no Python, no torch in the process -- and compared with the oracle: the counter streams against
engine (C), SMMC_FLAG_STREAM_REF against engine (R) (src/simulations.cpp:240-252)."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stock_market_monte_carlo_amd")
FLAG_STREAM_REF = 4


def fnv1a(a):
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a).view(np.uint8).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module")
def c_caller(tmp_path_factory, table):
    d = tmp_path_factory.mktemp("c_caller")
    exe = str(d / "c_caller")
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "tests", "c", "c_caller.c"), "-o", exe, "-L" + PKG, "-lsmmc_hip",
                          "-Wl,-rpath," + PKG], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    tab = str(d / "table.txt")
    with open(tab, "w") as f:
        for v in table:
            f.write(f"{float(v):.9g}\n")
    return exe, tab


def _run(c_caller, mode, n, p, seed, first, shards, flags=0):
    exe, tab = c_caller
    r = subprocess.run([exe, str(mode), str(n), str(p), str(seed), str(first), str(shards), str(flags), tab],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("mode_name,n,p,first,shards", [
    ("table", 100_003, 360, 0, 3),
    ("gaussian", 70_001, 360, (1 << 32) - 30_000, 2),   # global path ids cross 2^32 inside shard 0
    ("gaussian", 5, 1000, 7, 8),                          # fewer paths than shards: three shards are empty
    ("table", 0, 12, 0, 2),
])
def test_plain_c_program_against_the_oracle(c_caller, oracle, table, mode_name, n, p, first, shards):
    mode = oracle.MODE_GAUSSIAN if mode_name == "gaussian" else oracle.MODE_TABLE
    seed = 0x5EED5EED5EED5EED
    d = _run(c_caller, 1 if mode_name == "gaussian" else 0, n, p, seed, first, shards)
    o = oracle.counter_mc(oracle.make_params(mode, p, n, seed, first_path=first, table=table, n_bins=100, hist_lo=0.0,
                                             hist_hi=20000.0))
    st = o["stats"]
    base, extra = divmod(n, shards)
    assert d["group_size"] == shards and d["progress"] == n
    assert d["last_shard"] == [base * (shards - 1) + min(shards - 1, extra), base + (1 if shards - 1 < extra else 0)]
    assert d["final_fnv1a"] == fnv1a(o["final"])
    assert (d["count"], d["below"], d["underflow"], d["overflow"]) == (n, st.below, st.underflow, st.overflow)
    assert d["hist_fnv1a"] == fnv1a(o["hist"].astype(np.uint64)) == d["one_engine_hist_fnv1a"]
    assert (d["one_engine_count"], d["one_engine_below"]) == (n, st.below)
    if n:
        assert d["min_bits"] == int(np.float32(st.min).view(np.uint32)) and d["max_bits"] == int(np.float32(st.max).view(np.uint32))
        assert d["sum"] == pytest.approx(st.sum, rel=1e-12) and d["sumsq"] == pytest.approx(st.sumsq, rel=1e-12)
        assert d["one_engine_sum"] == pytest.approx(st.sum, rel=1e-12)
    else:
        assert d["min_bits"] == 0x7F800000 and d["max_bits"] == 0xFF800000 and d["sum"] == 0.0  # +inf / -inf
    assert d["rc_bad"] < 0


def test_plain_c_program_with_the_reference_stream(c_caller, oracle, table):
    n, p, seed, first = 50_001, 360, 123456789, 1000
    d = _run(c_caller, 0, n, p, seed, first, 3, FLAG_STREAM_REF)
    want, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, (seed + first) & 0xFFFFFFFF)
    assert d["final_fnv1a"] == fnv1a(want)
    assert d["count"] == n and d["below"] == int((want < np.float32(1000.0)).sum()) == d["one_engine_below"]
    assert d["sum"] == pytest.approx(float(want.astype(np.float64).sum()), rel=1e-12)
