"""benchmark_mc_* argument handling without a GPU: usage lines and exit codes follow the
reference (wrong argc -> usage text, exit 0: examples/benchmark_mc_gpu.cpp:56-61)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "stock_market_monte_carlo_amd", "bin")


@pytest.fixture(scope="module", autouse=True)
def built():
    from stock_market_monte_carlo_amd import build
    build.build_cli()


@pytest.mark.parametrize("prog,usage", [
    ("benchmark_mc_gpu", "usage: benchmark_mc_gpu <n_gpus> <n_months> <n_simulations>"),
    ("benchmark_mc_gpu_reduceBlock", "usage: benchmark_mc_gpu_reduceBlock <n_gpus> <n_months> <n_simulations>"),
    ("benchmark_mc_cpu_v2", "usage: visualize_returns <n_months> <n_simulations>"),
    ("benchmark_mc_cpu", "usage: visualize_returns <n_months> <n_simulations>"),
    ("benchmark_reduce_mean", "usage: compute_avg <n>"),
])
def test_usage_and_exit_code(prog, usage):
    r = subprocess.run([os.path.join(BIN, prog)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and usage in r.stdout and "argc: 1" in r.stdout


def test_no_gpu_is_a_loud_error_not_a_cpu_run():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([os.path.join(BIN, "benchmark_mc_gpu"), "1", "360", "1000"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and ("no HIP device visible" in r.stderr or "no MI355X visible" in r.stderr)
    assert "no CPU fallback" in r.stderr


def test_dropin_header_is_self_contained(tmp_path):
    """A caller written like the reference's examples (std::atomic<long>, vectors) compiles
    against the header alone -- the reference header needs fmt's transitive includes."""
    src = tmp_path / "caller.cpp"
    src.write_text('#include "stock_market_monte_carlo/simulations.h"\n'
                   "int main(){std::atomic<long> n{0}; std::vector<float> t{1.f}, out; "
                   "float (*f)(float,float) = &update_fund; (void)f; "
                   "void (*g)(std::atomic<long>&,long,int,float,std::vector<float>&,std::vector<float>&,int) = &mc_simulations_gpu; (void)g; "
                   "void (*h)(std::atomic<long>&,long,unsigned int,float,std::vector<float>&,std::vector<float>&) = &mc_simulations; (void)h; "
                   "return 0;}\n")
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)])


REF_DIR = os.path.join(ROOT, "oracle", "_ref")


@pytest.mark.skipif(not os.path.exists(os.path.join(REF_DIR, "benchmark_mc_gpu")),
                    reason="oracle/_ref not built (needs /root/reference at build time)")
def test_reference_gpu_main_starts_with_the_aliased_locale():
    """The reference's own benchmark_mc_gpu main, compiled untouched: without LOCPATH it dies in
    std::locale("en_US.UTF-8"); with the image's C.utf8 offered under that name it reaches the drop-in
    library (which, without a GPU, refuses loudly)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = os.path.join(REF_DIR, "benchmark_mc_gpu")
    env = {k: v for k, v in os.environ.items() if k != "LOCPATH"}
    r = subprocess.run([exe, "1", "360", "1000"], capture_output=True, text=True, cwd=ROOT, env=env)
    assert r.returncode != 0 and "locale" in r.stderr
    r = subprocess.run([exe, "1", "360", "1000"], capture_output=True, text=True, cwd=ROOT,
                       env=dict(env, LOCPATH=os.path.join(REF_DIR, "locale")))
    assert r.returncode != 0 and "no CPU fallback" in r.stderr and "locale" not in r.stderr
