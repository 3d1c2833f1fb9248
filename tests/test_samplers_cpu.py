"""SURVEY section 8 rows a5 / a6 pinned on the CPU: the drop-in's sample_returns_historical must draw
exactly the table entries the reference's loop draws (std::mt19937 + uniform_int_distribution<int>,
src/simulations.cpp:95-112) -- indices taken from the system-libstdc++ golden file -- and
sample_returns_gaussian (src/simulations.cpp:41-55) must be N(mean, std) to within sampling error."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stock_market_monte_carlo_amd")


@pytest.fixture(scope="module")
def host_check(tmp_path_factory):
    from stock_market_monte_carlo_amd import build
    build.build()
    exe = str(tmp_path_factory.mktemp("host") / "host_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "host_check.cpp"), "-o", exe, "-L" + PKG, "-lsmmc_hip",
                           "-Wl,-rpath," + PKG, "-pthread"])
    out = subprocess.check_output([exe], cwd=ROOT)
    return json.loads(out.decode().strip().splitlines()[-1])


def test_sample_returns_historical_draws_the_reference_indices(host_check, table):
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "libstdcxx_random.json")))
    assert host_check["table_len"] == table.size == golden["table_len"]
    seen = 0
    for u in golden["uniform_int"]:
        key = f"hist_{u['seed']}"
        if u["range"] != table.size or key not in host_check:
            continue
        want = table[np.array(u["out"], dtype=np.int64)]
        got = np.array(host_check[key], dtype=np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), key
        seen += 1
    assert seen == 4
    assert host_check["empty_threw"]


def test_sample_returns_historical_matches_the_oracle_index_map(host_check, table, oracle):
    # the C oracle's hand-written mt19937 + Lemire map (pinned to the same golden file) agrees too
    for seed in (0, 1000, 1001, 4294967295):
        idx = oracle.mt19937_indices(seed, table.size, 48)
        assert np.array_equal(np.array(host_check[f"hist_{seed}"], dtype=np.float32), table[idx])


def test_sample_returns_gaussian_moments(host_check):
    n, mean, std = host_check["gauss_n"], 0.5, 0.83333
    assert host_check["gauss_size"] == n and host_check["unseeded_size"] == 8 and host_check["gauss_repeat"]
    se_mean = std / np.sqrt(n)
    assert abs(host_check["gauss_mean"] - mean) < 5 * se_mean
    se_var = std ** 2 * np.sqrt(2.0 / n)
    assert abs(host_check["gauss_var"] - std ** 2) < 5 * se_var
    # normal shape: skewness 0 +- sqrt(6/n), excess kurtosis 0 +- sqrt(24/n)
    skew = host_check["gauss_m3"] / host_check["gauss_var"] ** 1.5
    kurt = host_check["gauss_m4"] / host_check["gauss_var"] ** 2 - 3.0
    assert abs(skew) < 5 * np.sqrt(6.0 / n) and abs(kurt) < 5 * np.sqrt(24.0 / n)


def test_scalar_functions(host_check, oracle):
    assert host_check["update_fund"] == 1005.0
    assert np.array_equal(np.array(host_check["mu"], dtype=np.float32), oracle.many_updates(1000.0, [1.0, -2.0, 3.5], 3))


def test_gaussian_cpu_reference_leg(oracle):
    """oracle/asref_cpu.cpp orc_asref_gaussian_mc -- BASELINE configs[0] as written (the reference's Gaussian
    demo path with a fixed seed; timed by bench.py's cpu_baseline): deterministic, independent of the
    thread count, and its final values have the log-normal law of 360 N(0.5 %, 0.83333 %) months."""
    a, used = oracle.asref_gaussian_mc(60000, 360, 1000.0, 0.5, 0.83333, 7, n_threads=1)
    b, used4 = oracle.asref_gaussian_mc(60000, 360, 1000.0, 0.5, 0.83333, 7, n_threads=4)
    assert used == 1 and used4 == 4 and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    c, _ = oracle.asref_gaussian_mc(60000, 360, 1000.0, 0.5, 0.83333, 8, n_threads=4)
    assert not np.array_equal(a, c) and np.array_equal(a[1:], c[:-1])  # path id of seed 8 is path id + 1 of seed 7
    # log final = sum of 360 log(1 + r/100), r ~ N(0.5, 0.83333): mean and variance by the delta method
    m, s = 0.005, 0.0083333
    mu = 360 * (np.log1p(m) - 0.5 * s * s / (1 + m) ** 2)
    sd = np.sqrt(360) * s / (1 + m)
    lg = np.log(a.astype(np.float64) / 1000.0)
    assert abs(lg.mean() - mu) < 5 * sd / np.sqrt(lg.size)
    assert abs(lg.std() / sd - 1.0) < 5 / np.sqrt(2 * lg.size)
    assert abs(a.astype(np.float64).mean() / (1000.0 * 1.005 ** 360) - 1.0) < 5 * 0.159 / np.sqrt(a.size)
