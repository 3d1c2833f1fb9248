"""Counts the instructions of the innermost period loop of a paths_kernel variant in a
gfx950 .s file (hipcc -S --cuda-device-only with the product's flags: `emit_asm()` below).

bench.py's `valu` object uses the plain VALU instruction count per path-period (every
instruction one 2-clock issue slot); tests/test_measurement_cpu.py asserts that the constants
there equal what this counts in the kernels as built.  The weighted figure (measured issue
costs relative to a plain VALU op, profiles/r01/ubench_instruction_rates.txt) is printed for
reading only: it over-counted in round 1 (fractions above 1) and is no longer reported.

usage: isa_loop_count.py [kernels.s] <variant> [periods]     variant e.g. ILi1ELi0ELb0E
"""
import os
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W = {'v_pk_fma_f32': 1.9, 'v_pk_mul_f32': 1.9, 'v_pk_add_f32': 1.9, 'v_mad_u64_u32': 2.29, 'v_mul_hi_u32': 1.78,
     'v_mul_lo_u32': 1.78, 'v_sqrt_f32_e32': 3.55, 'v_cvt_f32_u32_e32': 1.8, 'v_cvt_f32_i32_e32': 1.8,
     'v_rcp_f32_e32': 3.55, 'v_rsq_f32_e32': 3.55}
# paths_kernel<mode, div, dense>: (template arguments as mangled, periods per Philox block)
VARIANTS = {"gaussian": ("ILi1ELi0ELb0E", 4), "table": ("ILi0ELi0ELb1E", 8),
            "gaussian_checked": ("ILi1ELi2ELb0E", 4), "table_checked": ("ILi0ELi2ELb1E", 8)}


def emit_asm(out_path, source="smmc_kernels.hip"):
    """Device assembly of a kernel TU with exactly the product's compiler flags.  Kept beside the library's objects
    under the same content key (source + headers + flags + compiler: build._object_key), so that a second caller --
    the tests after a build, the driver's run after the builder's -- copies it instead of compiling for a minute."""
    import shutil
    sys.path.insert(0, ROOT)
    from stock_market_monte_carlo_amd import build as B
    cache = None
    try:
        cache = os.path.join(B.PKG, "_build", "asm", f"{source}.{B._object_key(os.path.join(B.CSRC, source))}.s")
        if os.path.exists(cache) and os.path.getsize(cache) > 0:
            shutil.copyfile(cache, out_path)
            return out_path
    except Exception:
        cache = None
    cmd = [B.hipcc()] + [f for f in B.FLAGS if f != "-fPIC"] + [
        "-x", "hip", "-S", "--cuda-device-only", "-Wno-unused-command-line-argument",
        "-I" + os.path.join(ROOT, "include"), "-I" + B.CSRC, "-o", out_path, os.path.join(B.CSRC, source)]
    subprocess.check_call(cmd)
    if cache:
        try:
            os.makedirs(os.path.dirname(cache), exist_ok=True)
            tmp = f"{cache}.{os.getpid()}.tmp"
            shutil.copyfile(out_path, tmp)
            os.replace(tmp, cache)
        except OSError:
            pass
    return out_path


def count(path, variant, kernel="paths_kernel"):
    """Counter of the opcodes of the innermost loop of smmc::(anon)::<kernel><variant>."""
    lines = open(path).read().splitlines()
    sym = f"_ZN4smmc12_GLOBAL__N_1{len(kernel)}{kernel}{variant}"
    beg = [i for i, l in enumerate(lines) if l.startswith(sym)][0]
    fin = [i for i, l in enumerate(lines) if i > beg and 's_endpgm' in l][0]
    body = lines[beg:fin]
    start = [i for i, l in enumerate(body) if 'Inner Loop Header: Depth=2' in l][0]
    end = [i for i, l in enumerate(body) if i > start and 's_cbranch_scc' in l][0]
    return Counter(l.split()[0] for l in body[start:end + 1] if l.strip() and l.strip()[0] not in ';.')


def kernel_opcodes(path, variant, kernel="paths_kernel"):
    """Mnemonics of every instruction of smmc::(anon)::<kernel><variant>, in program order."""
    lines = open(path).read().splitlines()
    sym = f"_ZN4smmc12_GLOBAL__N_1{len(kernel)}{kernel}{variant}"
    beg = [i for i, l in enumerate(lines) if l.startswith(sym)][0]
    fin = [i for i, l in enumerate(lines) if i > beg and 's_endpgm' in l][0]
    return [l.split()[0] for l in lines[beg + 1:fin + 1] if l.strip() and l.strip()[0] not in ';.' and not l.startswith('_Z')]


def fingerprint(path, variant, kernel="paths_kernel"):
    """sha256 over the kernel's instruction mnemonics in program order: changes whenever the compiled
    kernel gains, loses or reorders an instruction (register allocation alone does not change it).
    profiles/pmc_traffic.json stores it for the build that was profiled; tests/test_measurement_cpu.py
    compares it with the kernels as they compile now."""
    import hashlib
    return hashlib.sha256("\n".join(kernel_opcodes(path, variant, kernel)).encode()).hexdigest()


# what each workload key of profiles/pmc_traffic.json was measured on: (source file, kernel, template arguments)
TRAFFIC_KERNELS = {"gaussian": ("smmc_kernels.hip", "paths_kernel", "ILi1ELi0ELb0E"),
                   "table": ("smmc_kernels.hip", "paths_kernel", "ILi0ELi0ELb1E"),
                   "ref": ("smmc_ref_kernels.hip", "ref_windowed_kernel", "ILi0ELb0E"),
                   # the bundled table over 1000 periods cannot be proven safe for the fast divide: the checked variant runs
                   "ref_tree": ("smmc_ref_kernels.hip", "ref_tree_kernel", "ILi2ELb0ELi1077E")}
REF_WINDOWED_MAX = 454  # longer paths of the reference stream (up to 1816 periods) run ref_tree_kernel (two instantiations: <= 1077, <= 1816)


def traffic_kernel_of(key):
    """TRAFFIC_KERNELS entry name for a workload key `mode|paths|periods|outputs` of profiles/pmc_traffic.json."""
    mode, _, periods, _ = key.split("|")
    return "ref_tree" if mode == "ref" and int(periods) > REF_WINDOWED_MAX else mode


def source_digest(mode):
    """sha256 over the sources the profiled kernel of workload `mode` is compiled from, and the compiler
    flags: what bench.py can check at run time (no compiler needed) before it quotes a PMC figure."""
    import hashlib
    sys.path.insert(0, ROOT)
    from stock_market_monte_carlo_amd import build as B
    names = [TRAFFIC_KERNELS[mode][0], "smmc_device.h", "smmc_internal.h"]
    if TRAFFIC_KERNELS[mode][0] == "smmc_kernels.hip":
        names.append("smmc_bm_tables.inc")
    h = hashlib.sha256(" ".join(B.FLAGS).encode())
    for name in names:
        h.update(open(os.path.join(B.CSRC, name), "rb").read())
    return h.hexdigest()


def valu(c):
    return sum(n for k, n in c.items() if k.startswith('v_'))


def lds(c):
    return sum(n for k, n in c.items() if k.startswith('ds_'))


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0].endswith(".s"):
        path = args.pop(0)
    else:
        path = emit_asm("/tmp/smmc_kernels.s")
    variant = args[0]
    periods = int(args[1]) if len(args) > 1 else 4
    if variant in VARIANTS:
        variant, periods = VARIANTS[variant]
    c = count(path, variant)
    tot = sum(n * W.get(k, 1.0) for k, n in c.items() if k.startswith('v_') or k.startswith('ds_'))
    print(f"{variant}: {valu(c)} VALU + {lds(c)} LDS per {periods} periods "
          f"({valu(c) / periods:.2f} VALU/period); weighted {tot:.1f} -> {tot / periods:.1f} issue units per path-period")
    print("  ", dict(c.most_common(14)))
