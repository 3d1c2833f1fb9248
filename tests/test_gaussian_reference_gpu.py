"""BASELINE configs[0] / [1]'s Gaussian mode on the device against a reference-style CPU path.

The headline configuration is Gaussian, and bit-exactness vs oracle engine (C) only says that the kernel
computes what the build's own restatement computes (the two share the generated Box-Muller tables).  What
north_star asks for on top -- a "CPU-matching final-value distribution" -- is checked here against code that
shares nothing with the kernel: oracle/asref_cpu.cpp's orc_asref_gaussian_mc, i.e. libstdc++'s
std::default_random_engine + std::normal_distribution<float>(0.5, 0.83333) + update_fund as the reference's
own Gaussian sampler and compounding step are written (src/simulations.cpp:41-55, 14-16; parameters
examples/monte_carlo_simulated.cpp:11-12).  1e6 paths x 360 periods on either side; the statistics and their
limits are tests/two_sample.py's (log-value mean / std within 5 SE, seven quantiles within 5 SE, KS distance,
100-bucket histogram within sqrt(n) bands + chi-square of homogeneity); tests/test_gaussian_reference_cpu.py
shows the same comparison rejecting a standard deviation that is 1 % off."""
import os
import subprocess

import numpy as np
import pytest

import two_sample

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stock_market_monte_carlo_amd")
N, P, CAP, MEAN, STD = 1_000_000, 360, 1000.0, 0.5, 0.83333
BINS, LO, HI = 100, 0.0, 20000.0   # bench.py's histogram


@pytest.fixture(scope="module")
def cpu_reference(oracle):
    vals, _ = oracle.asref_gaussian_mc(N, P, CAP, MEAN, STD, 20260401)
    return vals, two_sample.product_histogram(vals, BINS, LO, HI)


@pytest.mark.parametrize("stream", [3, 2])
def test_device_gaussian_stream_matches_the_reference_style_cpu_path(cpu_reference, stream):
    import stock_market_monte_carlo_amd as S
    ref, (ref_hist, ref_under, ref_over) = cpu_reference
    eng = S.Engine(0)
    try:
        sim = S.Engine.make_sim(N, P, S.MODE_GAUSSIAN, 0x5EED5EED5EED5EED, initial_capital=CAP, gauss_mean=MEAN, gauss_std=STD,
                                n_bins=BINS, hist_lo=LO, hist_hi=HI, stream=stream)
        r = eng.simulate(sim, want_final=True, want_stats=True)
        st = eng.read_stats(r.stats_raw)
        got = r.final.cpu().numpy()
    finally:
        eng.close()
    # the fused histogram is the histogram of the values the kernel stored (the product's bucket contract)
    h, under, over = two_sample.product_histogram(got, BINS, LO, HI)
    assert np.array_equal(h, st.hist) and under == st.underflow and over == st.overflow and st.count == N
    fig = two_sample.compare(got, ref, CAP, hist_a=st.hist, hist_b=ref_hist)
    # count below the initial capital (examples/benchmark_mc_gpu.cpp:30-41): two binomial counts
    ref_below = int((ref < np.float32(CAP)).sum())
    assert abs(st.below - ref_below) <= 5 * np.sqrt(st.below + ref_below) + 5
    # mean of the final values themselves (what the CLI prints): the law's relative sd is 0.159
    assert abs(st.mean / float(ref.astype(np.float64).mean()) - 1.0) < 5 * 0.159 * np.sqrt(2.0 / N)
    print(f"stream v{stream}: log mean {fig['log_mean_se']:.2f} SE, log std {fig['log_std_se']:.2f} SE, KS {fig['ks_scaled']:.2f}, "
          f"chi2 {fig['hist_chi2'][0]:.1f} / {fig['hist_chi2'][1]}")


@pytest.mark.parametrize("n, periods, mean, std, hi", [
    (300_000, 1000, 0.5, 0.83333, 1.0e6),   # BASELINE configs[4]'s shape: 1000 periods (final values around 1.5e5)
    (500_000, 120, -0.2, 4.3, 5000.0),      # the spread of the historical table (4.3 % a month), a falling market
])
def test_other_shapes_and_parameters_match_the_reference_style_cpu_path(oracle, n, periods, mean, std, hi):
    """The same two-sample comparison away from the headline's parameters: more periods (the draw's error, if it had
    one, would accumulate over 1000 of them) and a standard deviation five times larger (the tails of the radius
    table carry more of the result)."""
    import stock_market_monte_carlo_amd as S
    ref, _ = oracle.asref_gaussian_mc(n, periods, CAP, mean, std, 20260402)
    ref_hist, _, _ = two_sample.product_histogram(ref, BINS, 0.0, hi)
    eng = S.Engine(0)
    try:
        sim = S.Engine.make_sim(n, periods, S.MODE_GAUSSIAN, 0xC0FFEE, initial_capital=CAP, gauss_mean=mean, gauss_std=std,
                                n_bins=BINS, hist_lo=0.0, hist_hi=hi)
        r = eng.simulate(sim, want_final=True, want_stats=True)
        st = eng.read_stats(r.stats_raw)
        got = r.final.cpu().numpy()
    finally:
        eng.close()
    h, under, over = two_sample.product_histogram(got, BINS, 0.0, hi)
    assert np.array_equal(h, st.hist) and under == st.underflow and over == st.overflow and st.count == n
    fig = two_sample.compare(got, ref, CAP, hist_a=st.hist, hist_b=ref_hist)
    print(f"{n} x {periods}, N({mean}, {std}): log mean {fig['log_mean_se']:.2f} SE, log std {fig['log_std_se']:.2f} SE, "
          f"KS {fig['ks_scaled']:.2f}, chi2 {fig['hist_chi2'][0]:.1f} / {fig['hist_chi2'][1]}")


def test_dropin_gaussian_matches_the_reference_style_cpu_path(cpu_reference, tmp_path):
    """smmc::mc_simulations_gpu_gaussian through the C++ drop-in layer (tests/cpp/dropin_check.cpp dumps its
    result vector): the same comparison, and the values are those of the C ABI's stream v3 for that seed."""
    import stock_market_monte_carlo_amd as S
    from stock_market_monte_carlo_amd import build
    build.build()
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "dropin_check.cpp"), "-o", exe, "-L" + PKG, "-lsmmc_hip",
                           "-Wl,-rpath," + PKG, "-pthread"])
    dump = tmp_path / "gauss.f32"
    subprocess.check_call([exe, str(N), str(P)], cwd=ROOT, env=dict(os.environ, SMMC_DROPIN_DUMP_GAUSS=str(dump)),
                          stdout=subprocess.DEVNULL)
    got = np.fromfile(dump, dtype=np.float32)
    assert got.size == N
    ref, (ref_hist, _, _) = cpu_reference
    h, _, _ = two_sample.product_histogram(got, BINS, LO, HI)
    two_sample.compare(got, ref, CAP, hist_a=h, hist_b=ref_hist)
    eng = S.Engine(0)
    try:
        sim = S.Engine.make_sim(N, P, S.MODE_GAUSSIAN, 4242, initial_capital=CAP, gauss_mean=MEAN, gauss_std=STD)  # dropin_check's seed
        same = eng.simulate(sim).final.cpu().numpy()
    finally:
        eng.close()
    assert np.array_equal(same.view(np.uint32), got.view(np.uint32))
