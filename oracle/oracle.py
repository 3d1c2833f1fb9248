"""ctypes loader for the CPU oracle (oracle/smmc_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsmmc_oracle.so")

MODE_TABLE = 0
MODE_GAUSSIAN = 1


class Params(C.Structure):
    _fields_ = [
        ("mode", C.c_int32),
        ("n_periods", C.c_uint32),
        ("seed", C.c_uint64),
        ("first_path", C.c_uint64),
        ("n_paths", C.c_uint64),
        ("initial_capital", C.c_float),
        ("gauss_mean", C.c_float),
        ("gauss_std", C.c_float),
        ("table", C.POINTER(C.c_float)),
        ("table_len", C.c_uint32),
        ("n_bins", C.c_uint32),
        ("hist_lo", C.c_float),
        ("hist_hi", C.c_float),
        ("below_threshold", C.c_float),
        ("stream", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("count", C.c_uint64),
        ("below", C.c_uint64),
        ("underflow", C.c_uint64),
        ("overflow", C.c_uint64),
        ("sum", C.c_double),
        ("sumsq", C.c_double),
        ("min", C.c_float),
        ("max", C.c_float),
    ]


def build(force=False):
    srcs = [os.path.join(_HERE, "smmc_oracle.c"), os.path.join(_HERE, "smmc_bm_tables.inc")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libsmmc_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_update_fund.restype = C.c_float
        L.orc_update_fund.argtypes = [C.c_float, C.c_float]
        L.orc_bm_radius.restype = C.c_float
        L.orc_bm_radius.argtypes = [C.c_uint32]
        L.orc_bm_radius_scan.restype = C.c_double
        L.orc_bm_radius_scan.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_bm3_radius.restype = C.c_float
        L.orc_bm3_radius.argtypes = [C.c_uint32]
        L.orc_bm3_radius_scan.restype = C.c_double
        L.orc_bm3_radius_scan.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_hist_bucket.restype = C.c_int32
        L.orc_hist_bucket.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint32]
        L.orc_div100_mismatches.restype = C.c_uint64
        L.orc_div100_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_ref_mc_simulations.restype = C.c_int
        L.orc_ref_mc_simulations.argtypes = [C.c_int64, C.c_uint32, C.c_float, C.c_void_p, C.c_uint32,
                                             C.c_uint32, C.c_void_p, C.c_int]
        L.orc_counter_mc.restype = C.c_int
        L.orc_counter_mc.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.POINTER(Stats),
                                     C.c_void_p, C.c_int]
        L.orc_counter_path_returns.argtypes = [C.POINTER(Params), C.c_uint64, C.c_void_p]
        L.orc_counter_path_indices.argtypes = [C.POINTER(Params), C.c_uint64, C.c_void_p]
        L.orc_draws_per_block.restype = C.c_uint32
        L.orc_draws_per_block.argtypes = [C.c_int32, C.c_uint32]
        L.orc_chunk_mean_var.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_values_stats.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_uint32, C.c_float, C.c_float,
                                       C.POINTER(Stats), C.c_void_p]
        L.orc_order_statistics.restype = C.c_int
        L.orc_order_statistics.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def update_fund(fund, ret):
    return float(lib().orc_update_fund(float(fund), float(ret)))


def many_updates(fund_value, returns, n_periods):
    r = _f32(returns)
    assert r.size >= n_periods
    out = np.empty(n_periods + 1, dtype=np.float32)
    out[0] = np.float32(fund_value)
    lib().orc_many_updates(r.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                           C.c_uint32(n_periods))
    return out


def mt19937_raw(seed, n):
    out = np.empty(n, dtype=np.uint32)
    lib().orc_mt19937_raw(C.c_uint32(seed), C.c_uint32(n), out.ctypes.data_as(C.c_void_p))
    return out


def mt19937_indices(seed, rng, n):
    out = np.empty(n, dtype=np.uint32)
    lib().orc_mt19937_indices(C.c_uint32(seed), C.c_uint32(rng), C.c_uint32(n),
                              out.ctypes.data_as(C.c_void_p))
    return out


def philox4x32_10(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_10(c.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p),
                            out.ctypes.data_as(C.c_void_p))
    return out


def box_muller_scaled(ua, ub, scale, shift):
    a = C.c_float()
    b = C.c_float()
    lib().orc_box_muller_scaled(C.c_uint32(ua), C.c_uint32(ub), C.c_float(scale), C.c_float(shift), C.byref(a), C.byref(b))
    return a.value, b.value


def box_muller(ua, ub):
    a = C.c_float()
    b = C.c_float()
    lib().orc_box_muller(C.c_uint32(ua), C.c_uint32(ub), C.byref(a), C.byref(b))
    return a.value, b.value


def box_muller3(ua, ub):
    """Counter stream v3's standard normals of (ua, ub)."""
    a = C.c_float()
    b = C.c_float()
    lib().orc_box_muller3(C.c_uint32(ua), C.c_uint32(ub), C.byref(a), C.byref(b))
    return a.value, b.value


def box_muller3_scaled(ua, ub, scale, shift):
    a = C.c_float()
    b = C.c_float()
    lib().orc_box_muller3_scaled(C.c_uint32(ua), C.c_uint32(ub), C.c_float(scale), C.c_float(shift), C.byref(a), C.byref(b))
    return a.value, b.value


def bm3_radius(ua):
    return float(lib().orc_bm3_radius(C.c_uint32(ua)))


def bm3_radius_scan(lo, hi, stride):
    return float(lib().orc_bm3_radius_scan(C.c_uint64(lo), C.c_uint64(hi), C.c_uint64(stride)))


def ref_mc_simulations(n_paths, n_periods, initial_capital, table, seed0, n_threads=0):
    """Engine (R): returns (final_values, threads_used)."""
    t = _f32(table)
    out = np.empty(n_paths, dtype=np.float32)
    used = lib().orc_ref_mc_simulations(n_paths, n_periods, float(initial_capital),
                                        t.ctypes.data_as(C.c_void_p), t.size, C.c_uint32(seed0 & 0xFFFFFFFF),
                                        out.ctypes.data_as(C.c_void_p), n_threads)
    return out, used


_asref = None


def _asref_lib():
    global _asref
    if _asref is None:
        so = os.path.join(_HERE, "libsmmc_asref.so")
        src = os.path.join(_HERE, "asref_cpu.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libsmmc_asref.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.orc_asref_mc_simulations.restype = C.c_int
        L.orc_asref_mc_simulations.argtypes = [C.c_int64, C.c_uint32, C.c_float, C.c_void_p, C.c_uint32, C.c_void_p,
                                               C.c_int, C.c_int, C.c_uint32]
        L.orc_asref_gaussian_mc.restype = C.c_int
        L.orc_asref_gaussian_mc.argtypes = [C.c_int64, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32,
                                            C.c_void_p, C.c_int]
        _asref = L
    return _asref


def asref_gaussian_mc(n_paths, n_periods, initial_capital, mean, std, seed0, n_threads=0):
    """BASELINE configs[0] as written: the reference's Gaussian demo path (std::default_random_engine +
    std::normal_distribution<float>, src/simulations.cpp:41-67) with a fixed seed, real libstdc++
    classes (oracle/asref_cpu.cpp).  Returns (final_values, threads_used)."""
    out = np.empty(n_paths, dtype=np.float32)
    used = _asref_lib().orc_asref_gaussian_mc(n_paths, n_periods, float(initial_capital), float(mean), float(std),
                                              C.c_uint32(seed0 & 0xFFFFFFFF), out.ctypes.data_as(C.c_void_p), n_threads)
    return out, used


def asref_mc_simulations(n_paths, n_periods, initial_capital, table, n_threads=0, fixed_seed0=None):
    """The reference's CPU loop as it really runs (src/simulations.cpp:240-252): a fresh
    std::random_device + mt19937 per path, real libstdc++ classes (oracle/asref_cpu.cpp).  Not
    reproducible unless fixed_seed0 is given, in which case it must equal engine (R).
    Returns (final_values, threads_used)."""
    _asref_lib()
    t = _f32(table)
    out = np.empty(n_paths, dtype=np.float32)
    used = _asref.orc_asref_mc_simulations(n_paths, n_periods, float(initial_capital), t.ctypes.data_as(C.c_void_p),
                                           t.size, out.ctypes.data_as(C.c_void_p), n_threads,
                                           0 if fixed_seed0 is None else 1,
                                           C.c_uint32((fixed_seed0 or 0) & 0xFFFFFFFF))
    return out, used


def make_params(mode, n_periods, n_paths, seed, first_path=0, initial_capital=1000.0, table=None,
                gauss_mean=0.5, gauss_std=0.83333, n_bins=0, hist_lo=0.0, hist_hi=1.0,
                below_threshold=1000.0, stream=3):
    p = Params()
    p.mode = mode
    p.n_periods = n_periods
    p.seed = seed
    p.first_path = first_path
    p.n_paths = n_paths
    p.initial_capital = initial_capital
    p.gauss_mean = gauss_mean
    p.gauss_std = gauss_std
    keep = None
    if table is not None:
        keep = _f32(table)
        p.table = keep.ctypes.data_as(C.POINTER(C.c_float))
        p.table_len = keep.size
    p.n_bins = n_bins
    p.hist_lo = hist_lo
    p.hist_hi = hist_hi
    p.below_threshold = below_threshold
    p.stream = stream
    p._keep = keep
    return p


def counter_mc(params, want_final=True, want_traj=False, n_threads=0):
    """Engine (C): returns dict(final, hist, stats, traj)."""
    n = int(params.n_paths)
    final = np.empty(n, dtype=np.float32) if want_final else None
    hist = np.zeros(max(int(params.n_bins), 1), dtype=np.uint64)
    traj = np.empty((n, params.n_periods + 1), dtype=np.float32) if want_traj else None
    st = Stats()
    rc = lib().orc_counter_mc(C.byref(params),
                              final.ctypes.data_as(C.c_void_p) if final is not None else None,
                              hist.ctypes.data_as(C.c_void_p) if params.n_bins else None,
                              C.byref(st),
                              traj.ctypes.data_as(C.c_void_p) if traj is not None else None, n_threads)
    if rc != 0:
        raise RuntimeError(f"orc_counter_mc failed: {rc}")
    return {"final": final, "hist": hist[: int(params.n_bins)], "stats": st, "traj": traj}


def counter_path_returns(params, path):
    out = np.empty(params.n_periods, dtype=np.float32)
    lib().orc_counter_path_returns(C.byref(params), C.c_uint64(path), out.ctypes.data_as(C.c_void_p))
    return out


def counter_path_indices(params, path):
    out = np.empty(params.n_periods, dtype=np.uint32)
    lib().orc_counter_path_indices(C.byref(params), C.c_uint64(path), out.ctypes.data_as(C.c_void_p))
    return out


def chunk_mean_var(values, chunk=256):
    v = _f32(values)
    nc = (v.size + chunk - 1) // chunk
    m = np.empty(nc, dtype=np.float32)
    q = np.empty(nc, dtype=np.float32)
    lib().orc_chunk_mean_var(v.ctypes.data_as(C.c_void_p), C.c_uint64(v.size), C.c_uint32(chunk),
                             m.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p))
    return m, q


def div100_mismatches(bits_lo, bits_hi):
    first = C.c_uint32(0)
    n = lib().orc_div100_mismatches(C.c_uint32(bits_lo), C.c_uint32(bits_hi), C.byref(first))
    return int(n), first.value


def bm_radius(ua):
    return float(lib().orc_bm_radius(C.c_uint32(ua)))


def bm_radius_scan(lo, hi, stride):
    return float(lib().orc_bm_radius_scan(C.c_uint64(lo), C.c_uint64(hi), C.c_uint64(stride)))


def values_stats(values, below_threshold=1000.0, n_bins=0, lo=0.0, hi=1.0):
    v = _f32(values)
    st = Stats()
    hist = np.zeros(max(n_bins, 1), dtype=np.uint64)
    lib().orc_values_stats(v.ctypes.data_as(C.c_void_p), v.size, below_threshold, n_bins, lo, hi, C.byref(st),
                           hist.ctypes.data_as(C.c_void_p) if n_bins else None)
    return st, hist[:n_bins]


def order_statistics(values, ranks):
    v = _f32(values)
    r = np.ascontiguousarray(ranks, dtype=np.uint64)
    out = np.empty(r.size, dtype=np.float32)
    rc = lib().orc_order_statistics(v.ctypes.data_as(C.c_void_p), v.size, r.ctypes.data_as(C.c_void_p), r.size,
                                    out.ctypes.data_as(C.c_void_p))
    if rc:
        raise RuntimeError("orc_order_statistics failed")
    return out


def quartiles(values):
    """examples/visualize_returns_cpu_v2.cpp:96-111: ranks 0, n/4, n/2, n/4 + n/2, n - 1."""
    n = len(values)
    q1, q2 = n // 4, n // 2
    return order_statistics(values, [0, q1, q2, min(q1 + q2, n - 1), n - 1])
