"""Pins against code compiled from the REFERENCE's own sources (oracle/_ref/, built by
`make -C oracle _ref`: src/helpers.cpp and the statistics helpers inside
examples/benchmark_mc_gpu.cpp, against this repository's header and library).  The
reference engine itself stays unbuildable; these are its host-side callers.

Runs on CPU.  Skipped when oracle/_ref/ has not been built (it needs /root/reference)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libsmmc_ref_callers.so")

pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def ref():
    from stock_market_monte_carlo_amd import _lib, build
    build.build()
    _lib.lib()  # loads torch's HIP runtime and libsmmc_hip.so first; the reference objects link to it
    L = C.CDLL(REF_SO)
    L.ref_update_mean_std.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ref_update_count_below_min.restype = C.c_long
    L.ref_update_count_below_min.argtypes = [C.c_void_p, C.c_long, C.c_float]
    L.ref_write_data_file.argtypes = [C.c_char_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long]
    L.ref_write_vector_file.argtypes = [C.c_char_p, C.c_void_p, C.c_long]
    return L


def test_oracle_statistics_match_reference_compiled_helpers(ref, oracle, table):
    """update_mean_std / update_count_below_min of examples/benchmark_mc_gpu.cpp:7-41 (compiled
    from the reference) on final values produced by the oracle engine."""
    final = oracle.counter_mc(oracle.make_params(oracle.MODE_TABLE, 120, 200000, 31, table=table))["final"]
    for v in (final, final[:1], final[:257], np.exp(np.random.default_rng(1).normal(7, 2, 50000)).astype(np.float32)):
        mean, std = C.c_float(), C.c_float()
        ref.ref_update_mean_std(v.ctypes.data_as(C.c_void_p), v.size, C.byref(mean), C.byref(std))
        st, _ = oracle.values_stats(v, below_threshold=1000.0)
        # the reference accumulates the sum in double in index order -> float(sum / n): same as the oracle
        assert np.float32(st.sum / v.size) == np.float32(mean.value)
        want_std = np.sqrt(max(st.sumsq / v.size - (st.sum / v.size) ** 2, 0.0))
        assert std.value == pytest.approx(want_std, rel=2e-4, abs=1e-3)  # reference: float residuals, two passes
        for thr in (1000.0, float(np.median(v)), 0.0, 1e30):
            st, _ = oracle.values_stats(v, below_threshold=thr)
            assert ref.ref_update_count_below_min(v.ctypes.data_as(C.c_void_p), v.size, thr) == st.below


def test_csv_writers_byte_identical_to_reference_helpers(ref, tmp_path):
    """write_data_file / write_vector_file (src/helpers.cpp:18-39, compiled from the reference)
    against the drop-in's, byte for byte."""
    exe = tmp_path / "writers_check"
    pkg = os.path.join(ROOT, "stock_market_monte_carlo_amd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "writers_check.cpp"), "-o", str(exe), "-L" + pkg,
                           "-lsmmc_hip", "-Wl,-rpath," + pkg, "-pthread"])
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        subprocess.check_call([str(exe)])
        r = np.array([1.5, -2.25, 0.1, 1e-7, 12345.678, -0.0], dtype=np.float32)
        v = np.array([1000.0, 1015.0, 992.1625, 993.154663, 1.0e9, 3.4e38, 1.17549435e-38], dtype=np.float32)
        ref.ref_write_data_file(b"theirs.csv", r.ctypes.data_as(C.c_void_p), r.size, v.ctypes.data_as(C.c_void_p), v.size)
        ref.ref_write_vector_file(b"outputs/theirs_vec.csv", v.ctypes.data_as(C.c_void_p), v.size)
        C.CDLL(None).fflush(None)  # the reference's writer announces its file through C stdio: not at interpreter exit
        assert open("outputs/ours.csv", "rb").read() == open("outputs/theirs.csv", "rb").read()
        assert open("outputs/ours_vec.csv", "rb").read() == open("outputs/theirs_vec.csv", "rb").read()
        assert open("outputs/ours.csv").read().startswith("Returns,,1.5,-2.25,0.1,1e-07,12345.7,-0,\nValues,1000,1015,")
    finally:
        os.chdir(cwd)
