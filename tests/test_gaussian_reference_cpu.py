"""The Gaussian mode's draw contract against a reference-style CPU path, on the CPU: oracle engine (C) --
the restatement the HIP kernels match bit for bit -- and oracle/asref_cpu.cpp's orc_asref_gaussian_mc
(std::default_random_engine + std::normal_distribution<float> + update_fund, src/simulations.cpp:41-55,
14-16, parameters examples/monte_carlo_simulated.cpp:11-12) are two samples of one law at BASELINE
configs[0]'s size.  The same comparison runs on the device in tests/test_gaussian_reference_gpu.py; here it
also proves that the comparison has the power to see a wrong draw."""
import numpy as np
import pytest

import two_sample

N, P, CAP, MEAN, STD = 1_000_000, 360, 1000.0, 0.5, 0.83333
BINS, LO, HI = 100, 0.0, 20000.0


@pytest.fixture(scope="module")
def cpu_reference(oracle):
    vals, _ = oracle.asref_gaussian_mc(N, P, CAP, MEAN, STD, 20260401)
    return vals, two_sample.product_histogram(vals, BINS, LO, HI)


@pytest.mark.parametrize("stream", [3, 2])
def test_oracle_gaussian_streams_match_the_reference_style_cpu_path(oracle, cpu_reference, stream):
    ref, (ref_hist, ref_under, ref_over) = cpu_reference
    o = oracle.counter_mc(oracle.make_params(oracle.MODE_GAUSSIAN, P, N, 0x5EED5EED5EED5EED, gauss_mean=MEAN, gauss_std=STD,
                                             n_bins=BINS, hist_lo=LO, hist_hi=HI, stream=stream))
    # the helper's histogram IS the product's contract: equal to the oracle's own buckets on the oracle's values
    h, under, over = two_sample.product_histogram(o["final"], BINS, LO, HI)
    assert np.array_equal(h, o["hist"]) and under == o["stats"].underflow and over == o["stats"].overflow
    r = two_sample.compare(o["final"], ref, CAP, hist_a=o["hist"], hist_b=ref_hist)
    assert abs(int(o["stats"].below) - int((ref < CAP).sum())) <= 5 * np.sqrt(2.0 * max((ref < CAP).sum(), 1)) + 5
    assert r["ks_scaled"] < two_sample.KS_LIMIT


@pytest.mark.parametrize("what,kw", [("std 1 % high", dict(gauss_std=STD * 1.01)), ("mean 0.002 high", dict(gauss_mean=MEAN + 0.002)),
                                     ("359 periods", dict(n_periods=P - 1))])
def test_the_comparison_sees_a_wrong_draw(oracle, cpu_reference, what, kw):
    """Power: a stream whose standard deviation is 1 % off, whose mean return is 0.002 percentage points off, or
    that compounds one period too few fails the comparison."""
    ref, (ref_hist, _, _) = cpu_reference
    args = dict(gauss_mean=MEAN, gauss_std=STD, n_periods=P)
    args.update(kw)
    o = oracle.counter_mc(oracle.make_params(oracle.MODE_GAUSSIAN, args.pop("n_periods"), N, 99, n_bins=BINS, hist_lo=LO, hist_hi=HI,
                                             **args))
    with pytest.raises(AssertionError):
        two_sample.compare(o["final"], ref, CAP, hist_a=o["hist"], hist_b=ref_hist)
