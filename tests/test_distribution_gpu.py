"""The draws themselves at a scale only the GPU reaches: 4e8 Gaussian draws against the normal
distribution's own bucket probabilities, 1e9 table draws against uniformity over the table, and the
independence of consecutive draws of a path.  Every value here is also covered bit for bit by the
oracle comparisons; what this adds is an independent statement about the distribution, from counts the
fused histogram kernel produced on the device."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SEED = 0x0DDBA11D15EA5E


def _phi(x):
    return 0.5 * (1.0 + math.erf(x / math.sqrt(2.0)))


@pytest.mark.parametrize("stream", [3, 2])
def test_gaussian_draws_follow_the_normal_distribution(stream):
    """One period: final = fl(fl(cap * a) / 100) with a = 100 + mean + std z, so the histogram of the
    final values is the histogram of z.  400 buckets over +-5 sigma plus the two tails, 4e8 draws:
    chi-square against the normal probabilities (std = 10 %: a bucket is 4e4 binary32 steps of the
    final value wide, so the roundings move 2e-5 of a bucket's mass), and the tail counts beyond
    5 sigma within 6 standard errors."""
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    try:
        n, cap, mean, std, bins = 400_000_000, 1000.0, 0.5, 10.0, 400
        centre, sigma = cap * (100.0 + mean) / 100.0, cap * std / 100.0
        lo, hi = centre - 5.0 * sigma, centre + 5.0 * sigma
        sim = S.Engine.make_sim(n, 1, S.MODE_GAUSSIAN, SEED, initial_capital=cap, gauss_mean=mean, gauss_std=std,
                                n_bins=bins, hist_lo=lo, hist_hi=hi, stream=stream)
        st = eng.read_stats(eng.simulate(sim, want_final=False, want_stats=True).stats_raw)
        assert st.count == n and int(st.hist.sum()) + st.underflow + st.overflow == n
        # the bucket edges as the kernel sees them: float32 lo / hi, double arithmetic in between
        lo32, hi32 = float(np.float32(lo)), float(np.float32(hi))
        edges = lo32 + (hi32 - lo32) * np.arange(bins + 1) / bins
        z = (edges - centre) / sigma
        p = np.diff([_phi(v) for v in z])
        expect = n * p
        chi2 = float((((st.hist.astype(np.float64) - expect) ** 2) / expect).sum())
        assert chi2 < bins + 6.0 * math.sqrt(2.0 * bins), chi2          # mean bins - 1, sd sqrt(2 bins)
        tail = n * _phi(-5.0)                                            # 114.7 draws expected per side
        assert abs(st.underflow - tail) < 6.0 * math.sqrt(tail) and abs(st.overflow - tail) < 6.0 * math.sqrt(tail)
        assert st.mean == pytest.approx(centre, abs=6.0 * sigma / math.sqrt(n))
        assert st.std == pytest.approx(sigma, rel=6.0 / math.sqrt(2.0 * n))
    finally:
        eng.close()


@pytest.mark.parametrize("t_len", [1127, 2048, 5000])
def test_table_draws_are_uniform_over_the_table(t_len):
    """A table whose entry i is the return i percent: one period turns the draw into cap (100 + i) / 100,
    one histogram bucket per entry.  1e9 draws (dense schedule: eight per Philox block; T = 5000: the
    sparse one): chi-square against uniformity."""
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    try:
        n = 1_000_000_000 if t_len <= 2048 else 500_000_000
        bins = min(t_len, 4096)
        width = -(-t_len // bins)                                        # entries per bucket
        eng.set_table(np.arange(t_len, dtype=np.float32))
        sim = S.Engine.make_sim(n, 1, S.MODE_TABLE, SEED, initial_capital=100.0, n_bins=bins, hist_lo=99.5,
                                hist_hi=99.5 + width * bins)
        st = eng.read_stats(eng.simulate(sim, want_final=False, want_stats=True).stats_raw)
        assert st.count == n and st.underflow == 0 and st.overflow == 0
        per_bucket = np.minimum(width * (np.arange(bins) + 1), t_len) - np.minimum(width * np.arange(bins), t_len)
        keep = per_bucket > 0
        expect = n * per_bucket[keep] / t_len
        assert int(st.hist[~keep].sum()) == 0
        chi2 = float((((st.hist[keep].astype(np.float64) - expect) ** 2) / expect).sum())
        dof = int(keep.sum()) - 1
        assert chi2 < dof + 6.0 * math.sqrt(2.0 * dof), (chi2, dof)
    finally:
        eng.close()


def test_consecutive_draws_of_a_path_are_uncorrelated():
    """Trajectories of 8 periods for 4e6 paths: the period returns (a - 100 from consecutive values) of
    one Philox block (4 draws: two Box-Muller pairs) and across the block boundary.  Correlation of every
    pair of periods within 5 / sqrt(n); so for their squares (a Box-Muller pair shares its radius: the
    angle has to decorrelate the magnitudes too)."""
    import stock_market_monte_carlo_amd as S
    eng = S.Engine(0)
    try:
        n, p = 4_000_000, 8
        sim = S.Engine.make_sim(n, p, S.MODE_GAUSSIAN, SEED, initial_capital=1000.0, gauss_mean=0.0, gauss_std=1.0)
        traj, _ = eng.simulate_keepdata(sim)
        t = traj.double()
        z = (t[:, 1:] / t[:, :-1] - 1.0) * 100.0                          # the returns, to 1e-5
        z = z - z.mean(dim=0, keepdim=True)
        zc = (z / z.std(dim=0, keepdim=True)).cpu().numpy()
        corr = zc.T @ zc / n
        sq = zc ** 2 - 1.0
        corr_sq = (sq.T @ sq / n) / 2.0                                   # var(z^2 - 1) = 2
        off = ~np.eye(p, dtype=bool)
        assert np.abs(corr[off]).max() < 5.0 / math.sqrt(n), np.abs(corr[off]).max()
        assert np.abs(corr_sq[off]).max() < 5.0 / math.sqrt(n), np.abs(corr_sq[off]).max()
        assert np.abs(np.diag(corr_sq) - 1.0).max() < 0.01               # kurtosis 3 within 1 %
    finally:
        eng.close()
