"""The reference's vector-add demo (src/gpu.cu:8-47, include/stock_market_monte_carlo/gpu.h), which north_star
names beside the engine: smmc_vector_add through the Python mirror against numpy's binary32 add, and the
reference's own examples/example_gpu.cpp compiled untouched against the drop-in header and library."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 255, 256, 1023, 1_000_000, (1 << 24) + 3])
def test_vector_add_equals_numpy_bit_for_bit(n):
    import stock_market_monte_carlo_amd as S
    rng = np.random.default_rng(n)
    a = rng.standard_normal(n).astype(np.float32) * np.float32(1e3)
    b = rng.standard_normal(n).astype(np.float32)
    if n > 8:
        a[:4] = [np.inf, -np.inf, 0.0, 3.0e38]
        b[:4] = [1.0, 1.0, -0.0, 3.0e38]  # inf, -inf, +0, overflow to inf
    out, seconds = S.vector_add_gpu(a, b)
    with np.errstate(over="ignore"):
        want = a + b
    assert out.shape == (n,) and np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert seconds >= 0.0 and (n == 0 or seconds > 0.0)


def test_vector_add_rejects_bad_arguments():
    import stock_market_monte_carlo_amd as S
    with pytest.raises(ValueError):
        S.vector_add_gpu(np.zeros(4, np.float32), np.zeros(5, np.float32))
    L = S._lib.lib()
    assert L.smmc_vector_add(None, None, None, 4, None) == -1 and b"NULL" in L.smmc_last_error()
    assert L.smmc_vector_add(None, None, None, -1, None) == -1
    assert L.smmc_vector_add(None, None, None, 0, None) == 0


def test_reference_example_gpu_runs_on_the_drop_in():
    """examples/example_gpu.cpp of the reference, compiled unmodified by `make -C oracle _ref` in the build container
    (oracle/_ref/ travels to the GPU box): vector_add then vector_add_gpu over N ones and twos, print_vector."""
    exe = os.path.join(ROOT, "oracle", "_ref", "example_gpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/example_gpu not built (needs /root/reference: make -C oracle _ref)")
    r = subprocess.run([exe, "1000"], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "CPU time:" in r.stdout and "GPU time:" in r.stdout
    vec = [l for l in r.stdout.splitlines() if l.startswith("v = [")]
    assert len(vec) == 1 and vec[0].split()[3:-1] == ["3.000"] * 1000
