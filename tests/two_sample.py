"""Two-sample comparison of final-value distributions (test helper, not a test module).

north_star asks for a "CPU-matching final-value distribution".  For the table mode the device runs the
reference CPU engine's own stream bit for bit (tests/test_ref_stream_gpu.py); the Gaussian mode has no such
stream in the reference -- its only Gaussian code is sample_returns_gaussian (src/simulations.cpp:41-55:
std::default_random_engine + std::normal_distribution<float>) -- so there the statement is distributional:
the device's final values and those of a reference-style CPU path (oracle/asref_cpu.cpp:
orc_asref_gaussian_mc, real libstdc++ classes + update_fund) are two samples of ONE law.

compare(a, b) checks that with statistics whose standard errors are known, each at a stated number of
standard errors, all in log space (log final value = a sum of 360 i.i.d. terms, normal to ~1e-3 in shape, so
its density at a quantile is well estimated by the pooled normal):

  * mean and standard deviation of the log final value (5 SE of the DIFFERENCE of two samples);
  * the quantiles {0.001, 0.01, 0.25, 0.5, 0.75, 0.99, 0.999}: SE of a sample quantile is
    sqrt(q (1 - q) / n) / f(x_q), the difference of two takes sqrt(2) of it (5 SE);
  * the two-sample Kolmogorov-Smirnov distance, D sqrt(n m / (n + m)) < 2.23 (p = 1e-4 asymptotically);
  * bucket counts of a histogram with the product's own contract: per bucket
    |a_k - b_k| <= 5 sqrt(a_k + b_k) + 5 (the difference of two independent counts has variance ~ a_k + b_k),
    and the chi-square of homogeneity over buckets with >= 20 pooled entries within 6 sigma of its mean.

Returns a dict of the figures it checked (for the test's failure message / a report).
"""
import math

import numpy as np

QUANTILES = (0.001, 0.01, 0.25, 0.5, 0.75, 0.99, 0.999)
KS_LIMIT = 2.23       # Kolmogorov distribution: P(K > 2.23) = 2 exp(-2 x 2.23^2) ~ 1e-4
N_SE = 5.0


def product_histogram(values, n_bins, lo, hi):
    """Bucket counts by the product's contract ("Histogram contract"), on the host:
    (counts, underflow, overflow)."""
    v = np.asarray(values, dtype=np.float32)
    lo32, hi32 = np.float32(lo), np.float32(hi)
    inv = float(n_bins) / (float(hi32) - float(lo32))
    under = int((v < lo32).sum())
    inside = v[(v >= lo32) & (v < hi32)]
    over = int(v.size - under - inside.size)
    b = np.minimum(((inside.astype(np.float64) - float(lo32)) * inv).astype(np.int64), n_bins - 1)
    return np.bincount(b, minlength=n_bins).astype(np.uint64), under, over


def compare(a, b, initial_capital=1000.0, hist_a=None, hist_b=None):
    """a, b: final values (float arrays) of two runs that should follow one law.  hist_a / hist_b: bucket
    counts of the two samples over the same buckets (optional).  Raises AssertionError with the figure that
    failed; returns the figures."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.size > 1000 and b.size > 1000 and (a > 0).all() and (b > 0).all()
    la, lb = np.log(a / initial_capital), np.log(b / initial_capital)
    n, m = la.size, lb.size
    pooled = np.concatenate([la, lb])
    mu, sd = float(pooled.mean()), float(pooled.std())
    out = {"n": (n, m), "log_mean": (float(la.mean()), float(lb.mean())), "log_std": (float(la.std()), float(lb.std()))}

    se_mean = sd * math.sqrt(1.0 / n + 1.0 / m)
    out["log_mean_se"] = abs(la.mean() - lb.mean()) / se_mean
    assert out["log_mean_se"] < N_SE, f"log-value means differ by {out['log_mean_se']:.2f} SE: {out['log_mean']}"
    se_std = sd * math.sqrt(0.5 / n + 0.5 / m)
    out["log_std_se"] = abs(la.std() - lb.std()) / se_std
    assert out["log_std_se"] < N_SE, f"log-value stds differ by {out['log_std_se']:.2f} SE: {out['log_std']}"

    sa, sb = np.sort(la), np.sort(lb)
    out["quantiles"] = {}
    for q in QUANTILES:
        xa, xb = float(sa[int(q * n)]), float(sb[int(q * m)])
        z = (0.5 * (xa + xb) - mu) / sd
        dens = math.exp(-0.5 * z * z) / (sd * math.sqrt(2.0 * math.pi))
        se = math.sqrt(q * (1.0 - q) * (1.0 / n + 1.0 / m)) / dens
        out["quantiles"][q] = (xa, xb, abs(xa - xb) / se)
        assert abs(xa - xb) < N_SE * se, f"quantile {q}: {xa} vs {xb}, {abs(xa - xb) / se:.2f} SE"

    # Kolmogorov-Smirnov: sup |F_a - F_b| over the pooled points
    grid = np.sort(pooled)
    d = float(np.abs(np.searchsorted(sa, grid, side="right") / n - np.searchsorted(sb, grid, side="right") / m).max())
    out["ks_d"] = d
    out["ks_scaled"] = d * math.sqrt(n * m / (n + m))
    assert out["ks_scaled"] < KS_LIMIT, f"KS distance {d:.3e}, scaled {out['ks_scaled']:.2f} >= {KS_LIMIT}"

    if hist_a is not None:
        ha, hb = np.asarray(hist_a, dtype=np.float64), np.asarray(hist_b, dtype=np.float64)
        assert ha.shape == hb.shape
        # the two samples may differ in size: scale b's counts to a's
        scale = n / m
        diff = np.abs(ha - hb * scale)
        band = N_SE * np.sqrt(ha + hb * scale * scale) + N_SE
        worst = int(np.argmax(diff - band))
        out["hist_worst_bucket"] = (worst, float(ha[worst]), float(hb[worst]))
        assert (diff <= band).all(), f"bucket {worst}: {ha[worst]:.0f} vs {hb[worst]:.0f} (scaled band {band[worst]:.1f})"
        keep = (ha + hb) >= 20
        tot = ha + hb
        ea, eb = tot * n / (n + m), tot * m / (n + m)
        chi2 = float(((ha[keep] - ea[keep]) ** 2 / ea[keep] + (hb[keep] - eb[keep]) ** 2 / eb[keep]).sum())
        dof = int(keep.sum()) - 1
        out["hist_chi2"] = (chi2, dof)
        assert chi2 < dof + 6.0 * math.sqrt(2.0 * dof), f"chi-square of homogeneity {chi2:.1f} with {dof} degrees of freedom"
    return out
