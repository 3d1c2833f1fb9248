"""HBM-resident statistics against the oracle:
order statistics bit-exact, integer counters and histogram exact, double sums to 1e-12."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(table):
    import stock_market_monte_carlo_amd as S
    e = S.Engine(0)
    e.set_table(table)
    yield e
    e.close()


def _dev(eng, a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(eng.tdevice)


def _cases():
    rng = np.random.default_rng(3)
    yield "lognormal", np.exp(rng.normal(8.5, 0.7, 100003)).astype(np.float32)
    yield "signed", rng.normal(0, 1e3, 65537).astype(np.float32)
    yield "ties", rng.integers(0, 7, 40000).astype(np.float32)
    yield "constant", np.full(5000, 1234.5, dtype=np.float32)
    yield "single", np.array([42.0], dtype=np.float32)
    yield "tiny", np.array([3.0, -1.0, 2.0], dtype=np.float32)
    yield "specials", np.array([0.0, -0.0, 1e-45, -1e-45, 3.4e38, -3.4e38, np.inf, -np.inf, 1.0, -1.0] * 13,
                               dtype=np.float32)
    yield "wide", (rng.normal(0, 1, 30011) * 10.0 ** rng.integers(-30, 30, 30011)).astype(np.float32)


@pytest.mark.parametrize("name,values", list(_cases()), ids=[c[0] for c in _cases()])
def test_order_statistics_bit_exact(eng, oracle, name, values):
    n = values.size
    ranks = sorted({0, n - 1, n // 2, n // 4, min(n // 4 + n // 2, n - 1), min(7, n - 1), n * 9 // 10, max(n - 2, 0)})[:8]
    got = eng.order_statistics(_dev(eng, values), ranks)
    want = oracle.order_statistics(values, ranks)
    if name == "specials":  # -0.0 and +0.0 tie under the oracle's comparison; as values they are equal
        assert np.array_equal(got, want), name
    else:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    if n >= 4:
        assert np.array_equal(eng.quartiles(_dev(eng, values)), oracle.quartiles(values))


def _octave_twins():
    """Two clusters an octave apart whose mantissas are equal down to bit 10: the ranks in the lower and in the upper
    cluster share the MIDDLE 11 bits of their keys and differ in the top 11 -- the one case in which pass 2's group
    table (keyed by the middle bits) cannot tell its groups apart and the kernel takes the select chain."""
    rng = np.random.default_rng(17)
    low_bits = rng.integers(0, 1024, 30001).astype(np.uint32)        # bits 0..9 vary, bits 10..22 are 1.5's
    a = (np.float32(1.5).view(np.uint32) | low_bits).view(np.float32)
    b = (np.float32(3.0).view(np.uint32) | low_bits[::-1]).view(np.float32)
    return np.concatenate([a, b, -a[:5000], np.float32(6.0) + np.zeros(3000, dtype=np.float32)])


@pytest.mark.parametrize("match", ["table", "chain"])
def test_order_statistics_group_match_forms_agree(oracle, table, match):
    """Round 4: passes 1 / 2 find a value's group through a 2048-entry LDS table instead of a select chain; a collision
    of two groups in the table (octave twins) falls back to the chain inside the kernel, and SMMC_RADIX_MATCH=chain
    forces the chain for every launch.  Every form, on every data set of this file and on the twins: the oracle's
    sort, bit for bit.  (The environment variable is read once per process: a child process per form.)"""
    import subprocess, sys, os, json
    code = r"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import stock_market_monte_carlo_amd as S
from oracle import oracle as O
import test_stats_gpu as T
e = S.Engine(0)
bad = []
cases = list(T._cases()) + [("octave_twins", T._octave_twins())]
for name, values in cases:
    n = values.size
    ranks = sorted({0, n - 1, n // 2, n // 4, min(n // 4 + n // 2, n - 1), min(7, n - 1), n * 9 // 10, max(n - 2, 0)})[:8]
    got = e.order_statistics(torch.from_numpy(values).to("cuda:0"), ranks)
    want = O.order_statistics(values, ranks)
    same = np.array_equal(got, want) if name == "specials" else np.array_equal(got.view(np.uint32), want.view(np.uint32))
    if not same: bad.append(name)
    if n >= 4 and not np.array_equal(e.quartiles(torch.from_numpy(values).to("cuda:0")), O.quartiles(values)): bad.append(name + " quartiles")
print(json.dumps({"bad": bad, "cases": len(cases)}))
"""
    env = dict(os.environ)
    env.pop("SMMC_RADIX_MATCH", None)
    if match == "chain":
        env["SMMC_RADIX_MATCH"] = "chain"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["bad"] == [] and out["cases"] == 9, out


def test_order_statistics_fuzz(eng, oracle):
    """Seeded random data sets and rank sets against the oracle's sort: sizes 1 ... 3e5, normal / log-normal / few distinct
    values / clusters an octave apart with shared mantissas (the group table's collision case) / signed mixtures, 1 ... 8
    ranks with repeats -- the number of groups per pass, their prefixes and the table-or-chain decision all vary."""
    rng = np.random.default_rng(20260406)
    for case in range(60):
        n = int(rng.choice([1, 2, 5, 63, 64, 65, 1000, 4097, 50001, 300007]))
        kind = case % 5
        if kind == 0:
            v = rng.normal(0, 10.0 ** rng.integers(-3, 6), n)
        elif kind == 1:
            v = np.exp(rng.normal(8.7, rng.uniform(0.01, 1.0), n))
        elif kind == 2:
            v = rng.choice(rng.normal(0, 100, 5), n)
        elif kind == 3:  # octave clusters: equal mantissa bits 10..22, different exponents
            mant = np.uint32(rng.integers(0, 1 << 13)) << np.uint32(10)
            exps = rng.integers(100, 150, 6).astype(np.uint32)
            bits = (rng.choice(exps, n).astype(np.uint32) << np.uint32(23)) | mant | rng.integers(0, 1024, n).astype(np.uint32)
            v = bits.view(np.float32)
        else:
            v = np.concatenate([rng.normal(-5, 1, n // 2), rng.normal(5e4, 3e3, n - n // 2)])
        v = np.ascontiguousarray(v, dtype=np.float32)
        k = int(rng.integers(1, 9))
        ranks = [int(r) for r in rng.integers(0, n, k)]
        got = eng.order_statistics(_dev(eng, v), ranks)
        want = oracle.order_statistics(v, ranks)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (case, n, kind, ranks)


def test_order_statistics_unaligned_views_and_ragged_sizes(eng, oracle):
    rng = np.random.default_rng(9)
    base = rng.normal(5000, 2000, 10007).astype(np.float32)
    d = _dev(eng, base)
    for off in (0, 1, 2, 3):
        for n in (1, 2, 3, 4, 5, 255, 257, 4099):
            v = d[off:off + n]
            got = eng.order_statistics(v.contiguous() if not v.is_contiguous() else v, [0, n // 2, n - 1])
            assert np.array_equal(got, oracle.order_statistics(base[off:off + n], [0, n // 2, n - 1])), (off, n)
            st = eng.read_stats(eng.values_stats(v, below_threshold=5000.0, n_bins=16, hist_lo=0.0, hist_hi=10000.0))
            ost, oh = oracle.values_stats(base[off:off + n], 5000.0, 16, 0.0, 10000.0)
            assert (st.count, st.below, st.underflow, st.overflow) == (ost.count, ost.below, ost.underflow, ost.overflow)
            assert np.array_equal(st.hist, oh) and st.min == ost.min and st.max == ost.max
            assert st.sum == pytest.approx(ost.sum, rel=1e-12)


def test_values_stats_matches_fused_simulation_statistics(eng, oracle, table):
    """The record computed from the final values equals the one the simulation kernel fuses."""
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE
    sim = Engine.make_sim(300007, 120, MODE_TABLE, 17, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
    r = eng.simulate(sim, want_stats=True)
    fused = eng.read_stats(r.stats_raw)
    st = eng.read_stats(eng.values_stats(r.final, below_threshold=1000.0, n_bins=100, hist_lo=0.0, hist_hi=20000.0))
    assert (st.count, st.below, st.underflow, st.overflow) == (fused.count, fused.below, fused.underflow, fused.overflow)
    assert np.array_equal(st.hist, fused.hist) and st.min == fused.min and st.max == fused.max
    assert st.sum == pytest.approx(fused.sum, rel=1e-12) and st.sumsq == pytest.approx(fused.sumsq, rel=1e-12)
    ost, oh = oracle.values_stats(r.final.cpu().numpy(), 1000.0, 100, 0.0, 20000.0)
    assert np.array_equal(st.hist, oh) and st.sum == pytest.approx(ost.sum, rel=1e-12)
    q = eng.quartiles(r.final)
    assert np.array_equal(q, oracle.quartiles(r.final.cpu().numpy()))


def test_bucket_accumulator_is_clean_between_calls_of_every_shape(eng, oracle, table):
    """Round 4: neither the record nor the engine's bucket accumulator is memset per call -- finalize_kernel folds the
    accumulator into the record and zeroes what it read.  One engine, calls of changing shape back to back: fused
    statistics of a simulation (one copy of the buckets), values_stats (sixteen), bucket counts from 0 to SMMC_MAX_BINS,
    empty inputs, and quartiles in between (its own self-cleaning histogram): every record equals the oracle's, i.e.
    nothing of an earlier call is left in a later one."""
    from stock_market_monte_carlo_amd import Engine, MODE_TABLE, MODE_GAUSSIAN
    rng = np.random.default_rng(5)
    values = np.exp(rng.normal(8.7, 0.4, 70001)).astype(np.float32)
    dv = _dev(eng, values)
    for i, bins in enumerate([100, 16, 1000, 7, 4096, 0, 100, 1, 4096, 3]):
        lo, hi = 0.0, float([20000.0, 9000.0, 30000.0][i % 3])
        st = eng.read_stats(eng.values_stats(dv, below_threshold=6000.0, n_bins=bins, hist_lo=lo, hist_hi=hi))
        ost, oh = oracle.values_stats(values, 6000.0, bins, lo, hi)
        assert (st.count, st.below, st.underflow, st.overflow) == (ost.count, ost.below, ost.underflow, ost.overflow), bins
        assert np.array_equal(st.hist, oh), bins
        mode = MODE_TABLE if i % 2 else MODE_GAUSSIAN
        sim = Engine.make_sim(20011 + 257 * i, 36, mode, 40 + i, n_bins=bins, hist_lo=lo, hist_hi=hi)
        r = eng.simulate(sim, want_stats=True)
        fused = eng.read_stats(r.stats_raw)
        fin = r.final.cpu().numpy()
        ost, oh = oracle.values_stats(fin, 1000.0, bins, lo, hi)
        assert (fused.count, fused.below, fused.underflow, fused.overflow) == (ost.count, ost.below, ost.underflow, ost.overflow), bins
        assert np.array_equal(fused.hist, oh), bins
        if i % 3 == 0:
            assert np.array_equal(eng.quartiles(r.final), oracle.quartiles(fin))
        empty = eng.read_stats(eng.values_stats(dv[:0], below_threshold=1.0, n_bins=bins, hist_lo=lo, hist_hi=hi))
        assert empty.count == 0 and int(np.asarray(empty.hist).sum()) == 0


def test_reference_named_helpers(oracle):
    """update_quartiles / update_mean_std / update_count_below_min / reduce_mean_gpu."""
    import stock_market_monte_carlo_amd as S
    rng = np.random.default_rng(21)
    v = np.exp(rng.normal(8, 1, 50001)).astype(np.float32)
    n_el = 40000  # the reference's helpers take a prefix length
    assert np.array_equal(S.update_quartiles(v, n_el), oracle.quartiles(v[:n_el]))
    mean, std = S.update_mean_std(v, n_el)
    d = v[:n_el].astype(np.float64)
    assert mean == pytest.approx(d.mean(), rel=1e-6) and std == pytest.approx(d.std(), rel=1e-5)
    assert S.update_count_below_min(3000.0, v, n_el) == int((v[:n_el] < 3000.0).sum())
    # examples/benchmark_reduce_mean.cpp: vec[i] = i, mean_cpu = float(sum) / n
    n = 3_000_017
    ramp = np.arange(n, dtype=np.float32)
    want = np.float32(np.float32(ramp.astype(np.float64).sum()) / np.float32(n))
    assert S.reduce_mean_gpu(ramp, n) == want
    assert S.reduce_mean_gpu(ramp, 10) == 4.5


def test_reduce_mean_streams_more_than_one_chunk(eng):
    n = (1 << 24) * 2 + 4321
    rng = np.random.default_rng(2)
    v = rng.uniform(0, 2, n).astype(np.float32)
    mean, total = eng.reduce_mean_host(v)
    assert total == pytest.approx(v.astype(np.float64).sum(), rel=1e-12)
    assert mean == np.float32(np.float32(total) / np.float32(n))


def test_full_size_quartiles_properties(eng):
    """1e8 final values: ranks are consistent with counting (no oracle at this size)."""
    from stock_market_monte_carlo_amd import Engine, MODE_GAUSSIAN
    n = 100_000_000
    r = eng.simulate(Engine.make_sim(n, 360, MODE_GAUSSIAN, 5))
    q = eng.quartiles(r.final)
    f = r.final
    assert float(f.min().item()) == q[0] and float(f.max().item()) == q[4]
    for rank, val in zip((n // 4, n // 2, n // 4 + n // 2), q[1:4]):
        below = int((f < float(val)).sum().item())
        at_or_below = int((f <= float(val)).sum().item())
        assert below <= rank < at_or_below
    assert q[0] <= q[1] <= q[2] <= q[3] <= q[4]


def test_bad_arguments(eng):
    from stock_market_monte_carlo_amd import SmmcError
    import torch
    v = torch.ones(10, device=eng.tdevice)
    with pytest.raises(SmmcError):
        eng.order_statistics(v, [10])
    with pytest.raises(SmmcError):
        eng.order_statistics(v, list(range(9)))
    with pytest.raises(SmmcError):
        eng.quartiles(v[:0])
    with pytest.raises(ValueError):
        eng.values_stats(v.double())
