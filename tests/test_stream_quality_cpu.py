"""Statistical sanity of counter stream v2 (oracle side; the HIP kernels are bit-identical to it):
normality of the Gaussian draws, independence along a path, across neighbouring path ids and
across seeds.  Loose bounds -- these catch wiring mistakes (reused counters, correlated lanes),
not subtle generator defects; Philox4x32-10 itself is pinned by its known-answer vectors."""
import numpy as np
import pytest
from scipy import stats


def _normals(oracle, n_paths, n_periods, seed, first=0):
    p = oracle.make_params(oracle.MODE_GAUSSIAN, n_periods, 1, seed, gauss_mean=0.0, gauss_std=1.0)
    return np.stack([oracle.counter_path_returns(p, first + i) for i in range(n_paths)]).astype(np.float64)


def test_gaussian_draws_are_standard_normal(oracle):
    z = _normals(oracle, 3000, 360, 2024)
    flat = z.ravel()
    assert abs(flat.mean()) < 4 / np.sqrt(flat.size)
    assert abs(flat.std() - 1) < 4 / np.sqrt(2 * flat.size)
    assert abs(stats.skew(flat)) < 0.01 and abs(stats.kurtosis(flat)) < 0.02
    assert stats.kstest(flat, "norm").pvalue > 1e-3
    # tails: P(|z| > 4) = 6.33e-5
    tail = (np.abs(flat) > 4).mean()
    assert 3e-5 < tail < 1.1e-4
    # the two outputs of one Box-Muller pair (cos / sin branch) and the two pairs of a block
    for a, b in ((0, 1), (0, 2), (1, 3), (3, 4)):
        assert abs(np.corrcoef(z[:, a::4].ravel()[:200000], z[:, b::4].ravel()[:200000])[0, 1]) < 0.01


def test_independence_along_and_across_paths(oracle, table):
    z = _normals(oracle, 2000, 64, 7)
    # lag-1..8 autocorrelation along a path
    for lag in range(1, 9):
        assert abs(np.corrcoef(z[:, :-lag].ravel(), z[:, lag:].ravel())[0, 1]) < 0.01
    # neighbouring path ids, same period (adjacent lanes of a wave)
    assert abs(np.corrcoef(z[:-1].ravel(), z[1:].ravel())[0, 1]) < 0.01
    # path ids that differ only in the high counter word
    hi = _normals(oracle, 500, 64, 7, first=1 << 32)
    assert abs(np.corrcoef(z[:500].ravel(), hi.ravel())[0, 1]) < 0.02
    # table mode: indices of neighbouring paths
    p = oracle.make_params(oracle.MODE_TABLE, 64, 1, 7, table=table)
    idx = np.stack([oracle.counter_path_indices(p, i) for i in range(2000)]).astype(np.float64)
    assert abs(np.corrcoef(idx[:-1].ravel(), idx[1:].ravel())[0, 1]) < 0.01
    for lag in range(1, 9):  # consecutive digits of one 64-bit word, and across words and blocks
        assert abs(np.corrcoef(idx[:, :-lag].ravel(), idx[:, lag:].ravel())[0, 1]) < 0.01


def test_seeds_and_modes_give_unrelated_streams(oracle, table):
    a = _normals(oracle, 300, 64, 1)
    b = _normals(oracle, 300, 64, 2)
    c = _normals(oracle, 300, 64, 1 << 32)  # differs in the key's high word only
    assert not np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(np.corrcoef(a.ravel(), b.ravel())[0, 1]) < 0.02 and abs(np.corrcoef(a.ravel(), c.ravel())[0, 1]) < 0.02
    assert np.array_equal(a, _normals(oracle, 300, 64, 1))  # same seed, same stream
    # the mode word separates the table stream from the Gaussian stream of the same path
    p = oracle.make_params(oracle.MODE_TABLE, 64, 1, 1, table=np.arange(1127, dtype=np.float32))
    t = np.stack([oracle.counter_path_returns(p, i) for i in range(300)]).astype(np.float64)
    assert abs(np.corrcoef(a.ravel(), t.ravel())[0, 1]) < 0.02


def test_final_value_distribution_matches_theory_in_gaussian_mode(oracle):
    """log(final / capital) = sum of 360 log(1 + r/100), r ~ N(0.5, 0.83333): compare the sample
    mean and standard deviation with the moments of one period's log-return."""
    n = 40000
    p = oracle.make_params(oracle.MODE_GAUSSIAN, 360, n, 99)
    fin = oracle.counter_mc(p)["final"].astype(np.float64)
    lr = np.log(fin / 1000.0)
    g = np.random.default_rng(0).normal(0.5, 0.83333, 4_000_000)
    one = np.log1p(g / 100.0)
    assert lr.mean() == pytest.approx(360 * one.mean(), abs=5 * np.sqrt(360) * one.std() / np.sqrt(n) + 1e-3)
    assert lr.std() == pytest.approx(np.sqrt(360) * one.std(), rel=0.02)
    assert stats.kstest((lr - lr.mean()) / lr.std(), "norm").pvalue > 1e-4
