"""The C++ drop-in layer (include/stock_market_monte_carlo/simulations.h) and the
benchmark_mc_* programs on the GPU, checked against the Python path and the oracle."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stock_market_monte_carlo_amd")


def fnv(a):
    h = 0xCBF29CE484222325
    for b in np.ascontiguousarray(a, dtype=np.float32).tobytes():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module")
def built():
    from stock_market_monte_carlo_amd import build
    build.build()
    build.build_cli()
    exe = os.path.join(ROOT, "tests", "cpp", "dropin_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "dropin_check.cpp"), "-o", exe, "-L" + PKG, "-lsmmc_hip",
                           "-Wl,-rpath," + PKG, "-pthread"])
    return exe


def test_cpp_dropin_matches_python_path_and_oracle(built, oracle, table):
    import stock_market_monte_carlo_amd as S
    n, p = 20000, 36
    # SMMC_DEVICE_MAP: the three shards of the n_gpus = 3 calls all run on this box's one GPU
    out = subprocess.check_output([built, str(n), str(p)], cwd=ROOT, env=dict(os.environ, SMMC_DEVICE_MAP="0,0,0"))
    d = json.loads(out.decode().strip().splitlines()[-1])
    # the multi-shard host path (thread fan-out, one engine per shard, host merge in shard order)
    assert d["multi_ran"] and d["multi_same"] and d["multi_summary_same"]
    assert d["multi_counter"] == d["n_multi"] and d["n_multi"] % 3 != 0
    # a polling thread sees a long run advance in steps (n/16-path chunks), never backwards
    assert d["progress_steps"] >= 4 and d["progress_monotone"]
    # one-pass variance at a large offset / on equal values (was 0.18 and NaN-prone)
    lv = (np.float32(1000.013) + np.float32(0.05) * np.sin(np.float32(0.001) * np.arange(1000000, dtype=np.float32))).astype(np.float32)
    assert d["lv_std"] == pytest.approx(float(lv.astype(np.float64).std()), rel=1e-4)
    assert d["lv_mean"] == pytest.approx(float(lv.astype(np.float64).mean()), rel=1e-7)
    # equal values: the double sums round at ~1e-16 relative, so up to ~1e-5 may be left; never NaN
    assert 0.0 <= d["eq_std"] < 2e-4 and d["eq_mean"] == pytest.approx(1000.013, rel=1e-7)
    want = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, p, n, 4242, table=table))["final"]
    assert d["gpu_hash"] == fnv(want) == d["cpu_hash"]            # both reference signatures, same stream
    assert d["gpu_hash"] == fnv(S.mc_simulations_gpu(n, p, 1000.0, table, seed=4242))
    assert d["counter_gpu"] == n and d["concurrent_ok"]
    cm, cv = oracle.chunk_mean_var(want)
    assert d["n_means"] == (n + 255) // 256
    assert d["mean0"] == pytest.approx(float(cm[0]), rel=1e-6) and d["var0"] == pytest.approx(float(cv[0]), rel=1e-5)
    assert d["threw"] and d["rows_ok"]
    assert d["keep_hash"] == fnv(want[:3000])
    assert np.array_equal(np.array(d["mu"], dtype=np.float32), oracle.many_updates(1000.0, [1.0, -2.0, 3.5], 3))
    assert d["mu_long_same"]
    assert d["update_fund"] == 1005.0
    g = oracle.counter_mc(oracle.make_params(oracle.MODE_GAUSSIAN, p, n, 4242, n_bins=50, hist_lo=0.0, hist_hi=5000.0))
    assert d["gauss_hash"] == fnv(g["final"])
    assert d["sum_count"] == n == d["hist_total"] and d["sum_below"] == g["stats"].below
    assert d["sum_mean"] == pytest.approx(g["stats"].sum / n, rel=1e-12)
    assert d["bundled"] == 1127 and d["sample_hist"] == 17 and d["sample_gauss"] == 9
    # statistics helpers (examples/visualize_returns_cpu_v2.cpp:83-138) and reduce_mean_gpu
    assert np.array_equal(np.array(d["quart"], dtype=np.float32), oracle.quartiles(want))
    w64 = want.astype(np.float64)
    assert d["hmean"] == pytest.approx(w64.mean(), rel=1e-6) and d["hstd"] == pytest.approx(w64.std(), rel=1e-5)
    assert d["hbelow"] == int((want < 1200.0).sum())
    ramp = np.arange(1000003, dtype=np.float32)
    assert np.float32(d["ramp_mean"]) == np.float32(np.float32(ramp.astype(np.float64).sum()) / np.float32(ramp.size))
    # CSV writers keep the reference's text format (src/helpers.cpp:18-39)
    assert open(os.path.join(ROOT, "outputs", "dropin_check.csv")).read() == "Returns,,1.5,-2.25,\nValues,1000,1015,992.162,"
    assert open(os.path.join(ROOT, "outputs", "dropin_check_vec.csv")).read() == "1000,1015,992.162,"


def _run(prog, *args, env=None):
    e = dict(os.environ, SMMC_SEED="99")
    e.update(env or {})
    return subprocess.run([os.path.join(PKG, "bin", prog), *map(str, args)], cwd=ROOT, env=e, capture_output=True,
                          text=True)


def test_benchmark_mc_gpu_cli_prints_reference_lines(built, oracle, table):
    r = _run("benchmark_mc_gpu", 1, 360, 200000)
    assert r.returncode == 0, r.stderr
    assert "n_periods: 360 | max_n_simulations: 200000" in r.stdout
    assert re.search(r"All 200000 simulation done in [0-9.e+-]+ s!", r.stdout)
    m = re.search(r"mean: ([0-9.]+) \| std: ([0-9.]+)", r.stdout)
    c = re.search(r"count_below 1000.0: ([0-9,]+) \(([0-9.]+)%\)", r.stdout)
    want = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 360, 200000, 99, table=table))["final"]
    mean = float(np.float32(want.astype(np.float64).sum() / want.size))
    assert float(m.group(1)) == pytest.approx(mean, abs=0.006)
    assert int(c.group(1).replace(",", "")) == int((want < 1000.0).sum())
    # Gaussian mode through the same program
    r = _run("benchmark_mc_gpu", 1, 360, 100000, env={"SMMC_MODE": "gaussian", "SMMC_JSON": "1"})
    assert r.returncode == 0 and '"program": "benchmark_mc_gpu"' in r.stdout


def test_benchmark_mc_gpu_three_shards_on_one_gpu(built):
    """`benchmark_mc_gpu 3 360 100003` (reference: n_gpus is just argv[1], examples/benchmark_mc_gpu.cpp:52)
    through the C++ multi-shard path; SMMC_DEVICE_MAP puts the three shards on the one GPU here."""
    one = _run("benchmark_mc_gpu", 1, 360, 100003, env={"SMMC_JSON": "1"})
    three = _run("benchmark_mc_gpu", 3, 360, 100003, env={"SMMC_DEVICE_MAP": "0,0,0", "SMMC_JSON": "1", "SMMC_VERBOSE": "1"})
    assert one.returncode == 0 and three.returncode == 0, three.stderr
    pick = lambda out: [l for l in out.splitlines() if l.startswith("mean:") or l.startswith("count_below")]  # noqa: E731
    assert pick(one.stdout) == pick(three.stdout) and len(pick(one.stdout)) == 2
    assert "All 100003 simulation done" in three.stdout
    assert three.stderr.count("smmc: shard") == 3 and "paths [66669, 100003)" in three.stderr
    # without the map, 3 shards on a 1-GPU box is the reference's error case (more GPUs than present)
    import torch
    if torch.cuda.device_count() < 3:
        r = _run("benchmark_mc_gpu", 3, 360, 1000)
        assert r.returncode == 1 and "exceeds the visible devices" in r.stderr


@pytest.mark.parametrize("periods", [360, 1000])
def test_benchmark_mc_gpu_eight_shards_as_configs_3_and_4_name_them(built, periods):
    """`benchmark_mc_gpu 8 <P> <N>`: the command line BASELINE configs[3] / configs[4] would be run with on an
    8-GPU node, all eight shards on this box's one GPU (eight engines, eight host threads, one process):
    mean and count as the one-shard run prints them, the remainder of N mod 8 kept (reference drops it,
    src/simulations.cu:602-603)."""
    n = 800005
    eight_map = ",".join(["0"] * 8)
    one = _run("benchmark_mc_gpu", 1, periods, n, env={"SMMC_JSON": "1"})
    eight = _run("benchmark_mc_gpu", 8, periods, n, env={"SMMC_DEVICE_MAP": eight_map, "SMMC_JSON": "1", "SMMC_VERBOSE": "1"})
    assert one.returncode == 0 and eight.returncode == 0, eight.stderr
    pick = lambda out: [l for l in out.splitlines() if l.startswith("mean:") or l.startswith("count_below")]  # noqa: E731
    assert pick(one.stdout) == pick(eight.stdout) and len(pick(one.stdout)) == 2
    assert f"All {n} simulation done" in eight.stdout
    assert eight.stderr.count("smmc: shard") == 8
    assert "paths [0, 100001)" in eight.stderr and f"paths [700005, {n})" in eight.stderr  # five shards carry one extra path


def test_python_mc_simulations_gpu_shards_concurrently(table, monkeypatch):
    import stock_market_monte_carlo_amd as S
    monkeypatch.setenv("SMMC_DEVICE_MAP", "0,0,0")
    n = 300007
    one = S.mc_simulations_gpu(n, 24, 1000.0, table, n_gpus=1, seed=5)
    three = S.mc_simulations_gpu(n, 24, 1000.0, table, n_gpus=3, seed=5)
    assert np.array_equal(one.view(np.uint32), three.view(np.uint32))
    monkeypatch.delenv("SMMC_DEVICE_MAP")
    import torch
    if torch.cuda.device_count() < 3:
        with pytest.raises(ValueError):
            S.mc_simulations_gpu(n, 24, 1000.0, table, n_gpus=3, seed=5)


def test_cpp_layer_stream_selection(built, oracle):
    """SMMC_STREAM=2 makes the C++ layer draw counter stream v2's Gaussians (the default, v3, is checked
    by hash in test_cpp_dropin_matches_python_path_and_oracle): every final value's bits, by hash."""
    n, p = 20000, 36
    out = subprocess.check_output([built, str(n), str(p)], cwd=ROOT, env=dict(os.environ, SMMC_STREAM="2"))
    d = json.loads(out.decode().strip().splitlines()[-1])
    g2 = oracle.counter_mc(oracle.make_params(oracle.MODE_GAUSSIAN, p, n, 4242, stream=2))["final"]
    g3 = oracle.counter_mc(oracle.make_params(oracle.MODE_GAUSSIAN, p, n, 4242))["final"]
    assert d["gauss_hash"] == fnv(g2) != fnv(g3)


def test_other_clis_run(built):
    r = _run("benchmark_mc_gpu_reduceBlock", 1, 360, 100000)
    assert r.returncode == 0 and "n_simulations: 100000" in r.stdout and "prob below min" in r.stdout
    assert "exact (fused on-device reduction" in r.stdout
    r = _run("benchmark_mc_gpu_reduceBlock", 2, 360, 1000)  # std::invalid_argument in the reference (simulations.cu:693)
    assert r.returncode == 1 and "only 1 GPU" in r.stderr
    r = _run("benchmark_mc_cpu_v2", 360, 100000)
    assert r.returncode == 0 and re.search(r"All 100000 simulation done in", r.stdout)
    r = _run("benchmark_mc_cpu", 360, 20000)
    assert r.returncode == 0 and re.search(r"All 20000 simulation done in", r.stdout)
    r = _run("benchmark_reduce_mean", 5000000)
    assert r.returncode == 0 and "mean_cpu: 2499999.50 | mean_gpu: 2499999.50" in r.stdout
    r = _run("benchmark_mc_gpu", 1, 360, 5000, env={"SMMC_TABLE": "/nonexistent.csv"})
    assert r.returncode == 0 and "bundled SYNTHETIC table (1127 entries)" in r.stdout


REF_DIR = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "benchmark_mc_cpu_v2")),
                    reason="oracle/_ref not built (needs /root/reference at build time)")
def test_reference_programs_compiled_unmodified_run_on_the_drop_in():
    """examples/benchmark_mc_cpu_v2.cpp and examples/benchmark_mc_cpu.cpp of the reference, compiled
    from its own sources (oracle/Makefile target _ref) against this header and library: the
    drop-in claim at link level."""
    env = dict(os.environ, SMMC_SEED="7")
    r = subprocess.run([os.path.join(REF_DIR, "benchmark_mc_cpu_v2"), "360", "2000000"], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "n_periods: 360 | max_n_simulations: 2000000" in r.stdout
    assert re.search(r"All 2000000 simulation done in [0-9.e+-]+ s!", r.stdout)
    r = subprocess.run([os.path.join(REF_DIR, "benchmark_mc_cpu"), "360", "100000"], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert re.search(r"All 100000 simulation done in [0-9.e+-]+ s!", r.stdout)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "benchmark_mc_gpu")),
                    reason="oracle/_ref not built (needs /root/reference at build time)")
def test_reference_gpu_programs_compiled_unmodified_run_on_the_drop_in(oracle, table):
    """examples/benchmark_mc_gpu.cpp and examples/benchmark_mc_gpu_reduceBlock.cpp of the reference,
    compiled untouched.  Their mains ask for the en_US.UTF-8 locale (benchmark_mc_gpu.cpp:45), which the
    image lacks: LOCPATH offers the image's own C.utf8 locale under that name (oracle/Makefile), nothing
    else changes.  The numbers they print must be the oracle's."""
    env = dict(os.environ, SMMC_SEED="99", LOCPATH=os.path.join(REF_DIR, "locale"))
    n = 200000
    r = subprocess.run([os.path.join(REF_DIR, "benchmark_mc_gpu"), "1", "360", str(n)], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert f"n_periods: 360 | max_n_simulations: {n}" in r.stdout
    assert re.search(rf"All {n} simulation done in [0-9.e+-]+ s!", r.stdout)
    want = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 360, n, 99, table=table))["final"]
    m = re.search(r"mean: ([0-9.]+) \| std: ([0-9.]+)", r.stdout)
    c = re.search(r"count_below 1000.0: ([0-9,]+) \(", r.stdout)
    w64 = want.astype(np.float64)
    assert float(m.group(1)) == pytest.approx(w64.mean(), abs=0.006)
    assert float(m.group(2)) == pytest.approx(w64.std(), rel=1e-4)
    assert int(c.group(1).replace(",", "")) == int((want < 1000.0).sum())
    # three shards through the reference's own main (its n_gpus is argv[1]), all on this box's GPU
    r3 = subprocess.run([os.path.join(REF_DIR, "benchmark_mc_gpu"), "3", "360", str(n)], cwd=ROOT,
                        env=dict(env, SMMC_DEVICE_MAP="0,0,0"), capture_output=True, text=True)
    assert r3.returncode == 0, r3.stderr
    assert re.search(r"mean: [0-9.]+ \| std: [0-9.]+", r3.stdout).group(0) == m.group(0)
    # the block-reduce program: per-block means merged its own way (examples/benchmark_mc_gpu_reduceBlock.cpp:7-26)
    rb = subprocess.run([os.path.join(REF_DIR, "benchmark_mc_gpu_reduceBlock"), "1", "360", str(n)], cwd=ROOT, env=env,
                        capture_output=True, text=True)
    assert rb.returncode == 0, rb.stderr
    mb = re.search(r"mean: ([0-9.]+) \| std: ([0-9.]+)", rb.stdout)
    cm, _ = oracle.chunk_mean_var(want)
    assert float(mb.group(1)) == pytest.approx(float(cm.astype(np.float64).mean()), rel=1e-4)


def test_reference_stream_through_the_drop_in(built, oracle, table):
    """SMMC_STREAM=ref: the reference-named entry points draw as the reference's CPU engine does
    (src/simulations.cpp:240-252: mt19937(seed + id), uniform_int_distribution, update_fund).  The C++
    check program (mc_simulations_gpu, mc_simulations, reduceBlock under fix_seed(4242)), the reference's
    own examples/benchmark_mc_gpu.cpp main and the Python mirror all give oracle engine (R)'s numbers."""
    import stock_market_monte_carlo_amd as S
    n, p = 20000, 36
    out = subprocess.check_output([built, str(n), str(p)], cwd=ROOT, env=dict(os.environ, SMMC_STREAM="ref"))
    d = json.loads(out.decode().strip().splitlines()[-1])
    want, _ = oracle.ref_mc_simulations(n, p, 1000.0, table, 4242)
    assert d["gpu_hash"] == fnv(want) == d["cpu_hash"] and d["concurrent_ok"]
    cm, cv = oracle.chunk_mean_var(want)
    assert d["mean0"] == pytest.approx(float(cm[0]), rel=1e-6) and d["var0"] == pytest.approx(float(cv[0]), rel=1e-5)
    assert np.array_equal(np.array(d["quart"], dtype=np.float32), oracle.quartiles(want))
    # mc_simulations_keepdata draws the same way (src/simulations.cpp:175-186): its final values are the same paths'
    assert d["rows_ok"] and d["keep_hash"] == fnv(want[:3000])
    # 1000 periods (BASELINE configs[4]'s length): the same entry points on ref_tree_kernel, keepdata included
    out = subprocess.check_output([built, "6000", "1000"], cwd=ROOT, env=dict(os.environ, SMMC_STREAM="ref"))
    dl = json.loads(out.decode().strip().splitlines()[-1])
    long_want, _ = oracle.ref_mc_simulations(6000, 1000, 1000.0, table, 4242)
    assert dl["gpu_hash"] == fnv(long_want) == dl["cpu_hash"] and dl["rows_ok"] and dl["keep_hash"] == fnv(long_want[:3000])
    # Python mirror
    got = S.mc_simulations(n, p, 1000.0, table, seed=4242, stream="ref")
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    got3 = S.mc_simulations_gpu(n + 1, p, 1000.0, table, n_gpus=1, seed=4241, stream="ref")
    assert np.array_equal(got3[1:].view(np.uint32), want.view(np.uint32))  # path id 1 of seed 4241 is path 0 of 4242
    # the benchmark program, at BASELINE configs[0] size: mean / std / count of the oracle's final values
    nb = 1_000_000
    r = _run("benchmark_mc_cpu_v2", 360, nb, env={"SMMC_STREAM": "ref", "SMMC_SEED": "1000", "SMMC_JSON": "1"})
    assert r.returncode == 0 and re.search(rf"All {nb} simulation done in", r.stdout), r.stderr
    if os.path.exists(os.path.join(REF_DIR, "benchmark_mc_gpu")):
        env = dict(os.environ, SMMC_SEED="1000", SMMC_STREAM="ref", LOCPATH=os.path.join(REF_DIR, "locale"))
        r = subprocess.run([os.path.join(REF_DIR, "benchmark_mc_gpu"), "1", "360", str(nb)], cwd=ROOT, env=env,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        big, _ = oracle.ref_mc_simulations(nb, 360, 1000.0, table, 1000)
        m = re.search(r"mean: ([0-9.]+) \| std: ([0-9.]+)", r.stdout)
        c = re.search(r"count_below 1000.0: ([0-9,]+) \(", r.stdout)
        w64 = big.astype(np.float64)
        assert float(m.group(1)) == pytest.approx(w64.mean(), abs=0.006)
        assert float(m.group(2)) == pytest.approx(w64.std(), rel=1e-4)
        assert int(c.group(1).replace(",", "")) == int((big < 1000.0).sum())


def test_drop_in_with_the_rccl_merge_gives_the_same_numbers(built):
    """SMMC_GROUP_MERGE=rccl: the C++ layer's n_gpus calls merge their per-device statistics records through
    the group's RCCL communicator (one rank on this box) instead of on the host; smmc::mc_summary's record and
    everything else the check program prints must not change."""
    n, p = 20000, 36
    runs = {}
    for merge in ("host", "rccl"):
        out = subprocess.check_output([built, str(n), str(p)], cwd=ROOT, env=dict(os.environ, SMMC_GROUP_MERGE=merge),
                                      timeout=600)
        runs[merge] = json.loads([l for l in out.decode().splitlines() if l.startswith("{")][-1])
    for key in ("gpu_hash", "cpu_hash", "gauss_hash", "keep_hash", "sum_count", "sum_mean", "sum_below", "hist_total", "quart",
                "hmean", "hstd", "hbelow"):
        assert runs["host"][key] == runs["rccl"][key], key
    assert runs["rccl"]["sum_count"] == n == runs["rccl"]["hist_total"]


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "benchmark_reduce_mean")),
                    reason="oracle/_ref not built (needs /root/reference at build time)")
def test_reference_reduce_mean_program_compiled_unmodified_runs_on_the_drop_in():
    """examples/benchmark_reduce_mean.cpp of the reference, compiled untouched against this
    header and library: its own CPU check and reduce_mean_gpu print the same mean."""
    env = dict(os.environ, LOCPATH=os.path.join(REF_DIR, "locale"))
    r = subprocess.run([os.path.join(REF_DIR, "benchmark_reduce_mean"), "5000000"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    m = re.search(r"mean_cpu: ([0-9.]+) \| mean_gpu: ([0-9.]+)", r.stdout)
    assert m and m.group(1) == m.group(2) == "2499999.50", r.stdout
