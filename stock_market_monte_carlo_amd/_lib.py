"""ctypes binding of the C ABI in include/smmc.h (libsmmc_hip.so).

There is no fallback: if the HIP library is missing or fails to load, importing the
engine raises.  Nothing here touches oracle/.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libsmmc_hip.so")
# Development only: tools/variant_build.sh writes experimental builds under _build/ (never over the
# product library) and the A/B tools point this at them.
if os.environ.get("SMMC_LIB"):
    import sys as _sys
    LIB_PATH = os.path.abspath(os.environ["SMMC_LIB"])
    print(f"stock_market_monte_carlo_amd: DEVELOPMENT library {LIB_PATH} (SMMC_LIB)", file=_sys.stderr)

ABI_VERSION = 4
MODE_TABLE = 0
MODE_GAUSSIAN = 1
FLAG_EXACT_DIV = 1
FLAG_STREAM_V2 = 2  # counter stream v2 (round 1's) instead of v3
FLAG_QUIET = 8
FLAG_HOST_NOPIN = 16  # simulate_to_host: never page-lock host_final for the call
FLAG_STREAM_REF = 4  # the reference CPU engine's own stream: per-path mt19937 + libstdc++ Lemire map (table mode)
CHUNK = 256
MAX_TABLE = 16384
MAX_BINS = 4096
MAX_RANKS = 8


class Sim(C.Structure):
    """smmc_sim"""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("mode", C.c_int32),
        ("seed", C.c_uint64),
        ("first_path", C.c_uint64),
        ("n_paths", C.c_uint64),
        ("n_periods", C.c_uint32),
        ("initial_capital", C.c_float),
        ("gauss_mean", C.c_float),
        ("gauss_std", C.c_float),
        ("n_bins", C.c_uint32),
        ("hist_lo", C.c_float),
        ("hist_hi", C.c_float),
        ("below_threshold", C.c_float),
        ("flags", C.c_uint32),
    ]


class Stats(C.Structure):
    """smmc_stats (header of the packed record; n_bins uint64 counts follow)"""
    _fields_ = [
        ("count", C.c_uint64),
        ("below", C.c_uint64),
        ("underflow", C.c_uint64),
        ("overflow", C.c_uint64),
        ("sum", C.c_double),
        ("sumsq", C.c_double),
        ("min", C.c_float),
        ("max", C.c_float),
        ("n_bins", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


# every symbol include/smmc.h declares: (name, restype, argtypes)
DIV_FAST, DIV_EXACT, DIV_CHECKED = 0, 1, 2  # smmc_engine_divide_kind
MERGE_HOST, MERGE_RCCL = 0, 1  # smmc_group_create

SYMBOLS = [
    ("smmc_update_fund", C.c_float, [C.c_float, C.c_float]),
    ("smmc_many_updates", None, [C.c_void_p, C.c_void_p, C.c_uint32]),
    ("smmc_abi_version", C.c_int, []),
    ("smmc_build_digest", C.c_char_p, []),
    ("smmc_vector_add", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_double)]),
    ("smmc_last_error", C.c_char_p, []),
    ("smmc_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("smmc_engine_create", C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    ("smmc_engine_destroy", None, [C.c_void_p]),
    ("smmc_engine_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("smmc_engine_get_stream", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    ("smmc_engine_wait_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("smmc_engine_release_to_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("smmc_engine_set_progress", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("smmc_engine_set_table", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    ("smmc_engine_simulate", C.c_int,
     [C.c_void_p, C.POINTER(Sim), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("smmc_engine_simulate_keepdata", C.c_int, [C.c_void_p, C.POINTER(Sim), C.c_void_p, C.c_void_p]),
    ("smmc_engine_sync", C.c_int, [C.c_void_p]),
    ("smmc_engine_simulate_to_host", C.c_int,
     [C.c_void_p, C.POINTER(Sim), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_void_p]),
    ("smmc_engine_prepare_host", C.c_int, [C.c_void_p, C.c_uint64]),
    ("smmc_host_register", C.c_int, [C.c_void_p, C.c_uint64]),
    ("smmc_host_unregister", C.c_int, [C.c_void_p]),
    ("smmc_group_prepare_host", C.c_int, [C.c_void_p, C.c_uint64]),
    ("smmc_engine_simulate_keepdata_to_host", C.c_int, [C.c_void_p, C.POINTER(Sim), C.c_void_p, C.c_void_p]),
    ("smmc_engine_values_stats", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_uint32, C.c_float, C.c_float, C.c_void_p]),
    ("smmc_engine_order_statistics", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("smmc_engine_quartiles", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("smmc_engine_reduce_mean_host", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    ("smmc_engine_host_values_summary", C.c_int,
     [C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_uint32, C.c_float, C.c_float, C.POINTER(Stats),
      C.c_void_p, C.c_void_p]),
    ("smmc_engine_timing", C.c_int, [C.c_void_p, C.c_int]),
    ("smmc_engine_kernel_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]),
    ("smmc_engine_kernel_clock", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("smmc_engine_selftest", C.c_int,
     [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("smmc_engine_geometry", C.c_int,
     [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("smmc_engine_divide_kind", C.c_int, [C.c_void_p, C.POINTER(Sim), C.c_int]),
    ("smmc_group_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    ("smmc_group_destroy", None, [C.c_void_p]),
    ("smmc_group_size", C.c_int, [C.c_void_p]),
    ("smmc_group_set_table", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    ("smmc_group_set_progress", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("smmc_group_simulate", C.c_int,
     [C.c_void_p, C.POINTER(Sim), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats), C.c_void_p]),
    ("smmc_group_shard", C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("smmc_group_device_record", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    ("smmc_group_timings", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("smmc_stats_bytes", C.c_uint64, [C.c_uint32]),
    ("smmc_stats_merge", C.c_int, [C.c_void_p, C.c_void_p]),
]

_lib = None


class SmmcError(RuntimeError):
    pass


def lib():
    """Loads libsmmc_hip.so once.  Raises if it is missing: there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SmmcError(
                f"{LIB_PATH} is missing: build it with `python -m stock_market_monte_carlo_amd.build` "
                "(hipcc, gfx950).  This package has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm wheels carry their own libamdhip64
        # (soname libamdhip64.so.7, the one our library asks for).  Loading torch first
        # makes the dynamic linker hand that copy to libsmmc_hip.so, so streams and
        # device pointers can cross between the two; the other order would map a
        # second runtime (/opt/rocm) that cannot see the device.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.smmc_abi_version() != ABI_VERSION:
            raise SmmcError(f"libsmmc_hip.so has ABI {L.smmc_abi_version()}, binding expects {ABI_VERSION}")
        _check_digest(L)
        _lib = L
    return _lib


def build_digest():
    """The digest the loaded library carries (smmc_build_digest)."""
    return lib().smmc_build_digest().decode()


def _check_digest(L):
    """The library must have been built from the sources that lie beside it: its embedded digest (flags + content
    of every source and header, build.source_digest()) against the tree's.  A stale product library is an error;
    a development library (SMMC_LIB) only a loud warning.  Without the sources (a copied .so) there is nothing
    to compare with and nothing is claimed."""
    import sys
    from . import build
    if not os.path.isdir(build.CSRC):
        return
    have, want = L.smmc_build_digest().decode(), build.source_digest()
    if have == want:
        return
    msg = (f"{LIB_PATH} was built from other sources or flags than this tree holds (library {have[:16]}..., sources "
           f"{want[:16]}...): rebuild it with `python -m stock_market_monte_carlo_amd.build`")
    if os.environ.get("SMMC_LIB"):
        print(f"stock_market_monte_carlo_amd: WARNING: {msg}", file=sys.stderr)
        return
    raise SmmcError("stale library: " + msg)


def check(rc):
    if rc != 0:
        raise SmmcError(f"smmc error {rc}: {lib().smmc_last_error().decode(errors='replace')}")
