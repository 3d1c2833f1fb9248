"""Development: builds an experimental VARIANT of the library from a patched scratch copy of csrc/ -- never from the
product sources, never over the product library -- for interleaved A/B runs on one box (SMMC_LIB=<variant>).

usage: [VARIANT_CSRC=dir] variant_build.py TAG [FILE 'OLD' 'NEW' ...]   ->  stock_market_monte_carlo_amd/_build/libsmmc_hip_TAG.so
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stock_market_monte_carlo_amd import build as B  # noqa: E402


def main():
    tag, edits = sys.argv[1], sys.argv[2:]
    assert len(edits) % 3 == 0
    tmp = tempfile.mkdtemp(prefix="smmc_variant_")
    csrc = os.path.join(tmp, "csrc")
    shutil.copytree(os.environ.get("VARIANT_CSRC", B.CSRC), csrc)  # VARIANT_CSRC: e.g. csrc/ of an earlier commit (git worktree)
    for i in range(0, len(edits), 3):
        path = os.path.join(csrc, edits[i])
        text = open(path).read()
        assert edits[i + 1] in text, f"{edits[i]}: pattern not found: {edits[i + 1][:60]}"
        open(path, "w").write(text.replace(edits[i + 1], edits[i + 2]))
    out = os.path.join(B.PKG, "_build", f"libsmmc_hip_{tag}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    stub = os.path.join(tmp, "digest.cpp")
    open(stub, "w").write('extern "C" __attribute__((visibility("default"))) const char *smmc_build_digest(void) { return "variant-%s"; }\n' % tag)
    objs = []
    procs = []
    for s in B.SOURCES:
        o = os.path.join(tmp, s + ".o")
        objs.append(o)
        procs.append(subprocess.Popen([B.hipcc()] + B.FLAGS + ["-x", "hip", "-c", "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "-o", o,
                                       os.path.join(csrc, s)]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call(["g++", "-O1", "-fPIC", "-c", stub, "-o", os.path.join(tmp, "digest.o")])
    subprocess.check_call([B.hipcc(), B.ARCH, "-shared", "-fPIC", "-o", out] + objs + [os.path.join(tmp, "digest.o"), "-ldl"])
    shutil.rmtree(tmp)
    print(out)


if __name__ == "__main__":
    main()
