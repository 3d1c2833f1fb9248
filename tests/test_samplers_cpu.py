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
