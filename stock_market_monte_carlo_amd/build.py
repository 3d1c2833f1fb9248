"""Builds libsmmc_hip.so (HIP kernels + C ABI + C++ drop-in layer) for gfx950, in-tree.

hipcc cross-compiles without a GPU.  The .so stays next to this file so that it
travels with the source tree (it is git-ignored, not pip-installed).

What ties the library to the sources it was built from is a CONTENT digest (`source_digest()`: every source,
every header, the compiler flags): the build embeds it in the library (`smmc_build_digest()`, include/smmc.h),
`stale()` compares the embedded one with the tree's, and the Python loader (_lib.py) refuses a library whose
digest differs from the sources beside it -- a forgotten rebuild cannot pass a test suite on old kernels.
"""
import hashlib
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libsmmc_hip.so")

SOURCES = ["smmc_kernels.hip", "smmc_ref_kernels.hip", "smmc_stats_kernels.hip", "smmc_vector_add.hip", "smmc_capi.cpp", "smmc_group.cpp", "smmc_dropin.cpp"]
HEADERS = [os.path.join(CSRC, "smmc_internal.h"), os.path.join(CSRC, "smmc_device.h"), os.path.join(CSRC, "smmc_host.h"),
           os.path.join(CSRC, "smmc_bm_tables.inc"),
           os.path.join(CSRC, "smmc_synthetic_table.inc"), os.path.join(ROOT, "include", "smmc.h"),
           os.path.join(ROOT, "include", "stock_market_monte_carlo", "simulations.h"),
           os.path.join(ROOT, "include", "stock_market_monte_carlo", "helpers.h"),
           os.path.join(ROOT, "include", "stock_market_monte_carlo", "gpu.h")]

# -ffp-contract=off: results must be bit-identical to the CPU oracle; every FMA in
# the sources is explicit.  Correctly rounded fp32 divide/sqrt is hipcc's default
# and must stay on.
# -fno-slp-vectorize: v_pk_fma_f32 issues at the rate of two scalar FMAs on gfx950, and the
# register-pair moves SLP packing adds are pure overhead in the VALU-bound loops (131 -> 129
# instructions per 4 Gaussian periods).
ARCH = "--offload-arch=gfx950"
FLAGS = ["-O3", ARCH, "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize",
         "-Wall", "-Wextra", "-Wno-unused-parameter"]
DIGEST_MARK = "SMMC_BUILD_DIGEST="  # the embedded string: DIGEST_MARK + 64 hex digits (also found by scanning the file)


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the MI355X engine cannot be built")
    return exe


def _files():
    return [os.path.join(CSRC, s) for s in SOURCES] + HEADERS


def source_digest():
    """sha256 over the compiler flags and the CONTENT of every source and header the library is built from
    (names relative to the repository, so that a copy of the tree has the same digest)."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for path in sorted(_files()):
        if not os.path.exists(path):
            continue
        h.update(b"\0" + os.path.relpath(path, ROOT).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def embedded_digest(lib_path=LIB):
    """The digest a built library carries, read from the file (no dlopen); None if there is none."""
    try:
        with open(lib_path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    at = blob.find(DIGEST_MARK.encode())
    if at < 0:
        return None
    hexd = blob[at + len(DIGEST_MARK):at + len(DIGEST_MARK) + 64]
    try:
        return hexd.decode("ascii") if len(hexd) == 64 and int(hexd, 16) >= 0 else None
    except ValueError:
        return None


def stale():
    """True when there is no library or it was built from other sources / flags than the tree holds now."""
    return embedded_digest() != source_digest()


OBJ = os.path.join(PKG, "_build", "obj")  # git-ignored object cache


_toolchain = None


def _toolchain_id():
    """hipcc's own version line: part of every object's key (a new compiler must not link old objects)."""
    global _toolchain
    if _toolchain is None:
        try:
            _toolchain = subprocess.run([hipcc(), "--version"], capture_output=True, text=True, timeout=120).stdout.strip()
        except (OSError, subprocess.SubprocessError):
            _toolchain = "unknown"
    return _toolchain


def _object_key(src):
    """What an object file depends on: its source, every header (one list for all translation units), the
    flags, the compiler.  Objects are reused exactly while this is unchanged."""
    h = hashlib.sha256((" ".join(FLAGS) + "\n" + _toolchain_id()).encode())
    for path in [src] + sorted(p for p in HEADERS if os.path.exists(p)):
        with open(path, "rb") as fh:
            h.update(b"\0" + fh.read())
    return h.hexdigest()[:16]


def _compile(src, verbose):
    base = os.path.basename(src)
    obj = os.path.join(OBJ, f"{base}.{_object_key(src)}.o")
    if os.path.exists(obj):
        return obj
    for name in os.listdir(OBJ):  # objects of this source under other keys
        if name.startswith(base + ".") and name.endswith(".o"):
            os.remove(os.path.join(OBJ, name))
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


def _digest_object(digest, verbose):
    """The translation unit that carries the digest: smmc_build_digest() (include/smmc.h)."""
    src = os.path.join(OBJ, "smmc_build_digest.cpp")
    obj = os.path.join(OBJ, "smmc_build_digest.o")
    text = ('// generated by stock_market_monte_carlo_amd/build.py\n'
            'extern "C" __attribute__((visibility("default"))) const char *smmc_build_digest(void) {\n'
            f'  static const char text[] = "{DIGEST_MARK}{digest}";\n'
            f'  return text + {len(DIGEST_MARK)};\n'
            '}\n')
    with open(src, "w") as fh:
        fh.write(text)
    cmd = ["g++", "-O1", "-fPIC", "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    digest = source_digest()  # of what is about to be compiled
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for name in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, name))
    # one hipcc per translation unit, side by side (the reference-stream kernels alone take 40 s)
    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda src: _compile(src, verbose), srcs))
    objs.append(_digest_object(digest, verbose))
    if source_digest() != digest:
        raise RuntimeError("a source changed while the library was being built: run the build again")
    # link beside the target and rename over it: a process that has the old library mapped
    # keeps its inode instead of seeing the file truncated under it
    tmp = LIB + ".tmp%d" % os.getpid()
    cmd = [hipcc(), ARCH, "-shared", "-fPIC", "-o", tmp] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    try:
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    if embedded_digest() != digest:
        raise RuntimeError("the linked library does not carry the digest it was built with")
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
