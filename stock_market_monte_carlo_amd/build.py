"""Builds libsmmc_hip.so (HIP kernels + C ABI + C++ drop-in layer) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  The .so stays next to this file so that it
travels with the source tree (it is git-ignored, not pip-installed).
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libsmmc_hip.so")

SOURCES = ["smmc_kernels.hip", "smmc_ref_kernels.hip", "smmc_stats_kernels.hip", "smmc_vector_add.hip", "smmc_capi.cpp", "smmc_group.cpp", "smmc_dropin.cpp"]
HEADERS = [os.path.join(CSRC, "smmc_internal.h"), os.path.join(CSRC, "smmc_device.h"), os.path.join(CSRC, "smmc_bm_tables.inc"),
           os.path.join(CSRC, "smmc_synthetic_table.inc"), os.path.join(ROOT, "include", "smmc.h"),
           os.path.join(ROOT, "include", "stock_market_monte_carlo", "simulations.h")]

# -ffp-contract=off: results must be bit-identical to the CPU oracle; every FMA in
# the sources is explicit.  Correctly rounded fp32 divide/sqrt is hipcc's default
# and must stay on.
# -fno-slp-vectorize: v_pk_fma_f32 issues at the rate of two scalar FMAs on gfx950, and the
# register-pair moves SLP packing adds are pure overhead in the VALU-bound loops (131 -> 129
# instructions per 4 Gaussian periods).
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize",
         "-Wall", "-Wextra", "-Wno-unused-parameter"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the MI355X engine cannot be built")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


OBJ = os.path.join(PKG, "_build", "obj")  # git-ignored; objects are reused while their source and the headers are older


def _compile(src, verbose):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    newest = max(os.path.getmtime(d) for d in [src] + [h for h in HEADERS if os.path.exists(h)])
    if os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj
    tmp = obj + ".tmp%d" % os.getpid()
    cmd = [hipcc()] + FLAGS + ["-x", "hip", "-c", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", tmp, src]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, obj)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return obj


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for name in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, name))
    # one hipcc per translation unit, side by side (the reference-stream kernels alone take 40 s)
    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda src: _compile(src, verbose), srcs))
    # link beside the target and rename over it: a process that has the old library mapped
    # keeps its inode instead of seeing the file truncated under it
    tmp = LIB + ".tmp%d" % os.getpid()
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


def build_cli(force=False, verbose=False):
    """The benchmark_mc_* command-line programs (reference examples/benchmark_mc_*.cpp)."""
    out_dir = os.path.join(PKG, "bin")
    os.makedirs(out_dir, exist_ok=True)
    build(force=force, verbose=verbose)
    cli_dir = os.path.join(CSRC, "cli")
    built = []
    for name in sorted(os.listdir(cli_dir)) if os.path.isdir(cli_dir) else []:
        if not name.endswith(".cpp"):
            continue
        src = os.path.join(cli_dir, name)
        exe = os.path.join(out_dir, name[:-4])
        if force or not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(LIB)):
            tmp = exe + ".tmp%d" % os.getpid()
            cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + cli_dir, src, "-o", tmp,
                   "-L" + PKG, "-lsmmc_hip", "-Wl,-rpath,$ORIGIN/..", "-pthread"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            os.replace(tmp, exe)
        built.append(exe)
    return built


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    for exe in build_cli(verbose=True):
        print(exe)
