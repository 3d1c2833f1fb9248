"""The library that gets loaded is the library of THESE sources: libsmmc_hip.so carries a
content digest of its sources, headers and compiler flags (smmc_build_digest, include/smmc.h); build.stale()
compares digests, not modification times; the loader refuses a library that does not match the tree."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stock_market_monte_carlo_amd")


def test_the_built_library_carries_the_digest_of_the_tree():
    from stock_market_monte_carlo_amd import _lib, build
    assert not build.stale()
    d = build.source_digest()
    assert len(d) == 64 and build.embedded_digest() == d == _lib.build_digest()
    # the flags are part of it (a flag such as -ffp-contract must not leave the library "fresh")
    saved = list(build.FLAGS)
    try:
        build.FLAGS.remove("-ffp-contract=off")
        assert build.source_digest() != d and build.stale()
    finally:
        build.FLAGS[:] = saved
    assert not build.stale()
    # every file a translation unit includes from this repository is in the digest
    import re
    listed = {os.path.basename(p) for p in build.HEADERS}
    for src in build.SOURCES + [os.path.basename(h) for h in build.HEADERS if h.endswith(".h")]:
        path = os.path.join(build.CSRC, src)
        if not os.path.exists(path):
            continue
        for inc in re.findall(r'#include "([^"]+)"', open(path).read()):
            assert os.path.basename(inc) in listed, f"{src} includes {inc}, which build.HEADERS does not list"


def _import_in(tree, env=None):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from stock_market_monte_carlo_amd import _lib, build\n"
            "L = _lib.lib(); print('LOADED', build.stale())\n" % tree)
    e = dict(os.environ)
    e.pop("SMMC_LIB", None)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, cwd=tree)


def test_a_changed_source_without_a_rebuild_fails_at_import(tmp_path):
    tree = tmp_path / "tree"
    shutil.copytree(PKG, tree / "stock_market_monte_carlo_amd", ignore=shutil.ignore_patterns("_build", "bin", "__pycache__"))
    shutil.copytree(os.path.join(ROOT, "include"), tree / "include")
    ok = _import_in(str(tree))
    assert ok.returncode == 0 and "LOADED False" in ok.stdout, ok.stderr
    header = tree / "stock_market_monte_carlo_amd" / "csrc" / "smmc_device.h"
    header.write_text(header.read_text() + "\n// an edit that was never compiled\n")
    bad = _import_in(str(tree))
    assert bad.returncode != 0 and "stale library" in bad.stderr and "rebuild it" in bad.stderr, bad.stderr
    # a development library (SMMC_LIB) is loaded with a loud warning instead
    dev = _import_in(str(tree), env={"SMMC_LIB": str(tree / "stock_market_monte_carlo_amd" / "libsmmc_hip.so")})
    assert dev.returncode == 0 and "LOADED True" in dev.stdout and "WARNING" in dev.stderr, dev.stderr
    # touching a file (new mtime, same bytes) does not make anything stale
    header.write_text(header.read_text().replace("\n// an edit that was never compiled\n", ""))
    os.utime(header, None)
    again = _import_in(str(tree))
    assert again.returncode == 0 and "LOADED False" in again.stdout, again.stderr
