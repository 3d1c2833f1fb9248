"""CPU-side checks of the product library: it loads, exports every symbol
include/smmc.h declares, the host scalar functions match the oracle, and the engine
refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from stock_market_monte_carlo_amd import build
    build.build()
    from stock_market_monte_carlo_amd import _lib
    return _lib.lib()


def test_every_declared_symbol_is_exported(L):
    hdr = open(os.path.join(ROOT, "include", "smmc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(smmc_[a-z_0-9]+)\s*\(", hdr))
    from stock_market_monte_carlo_amd import _lib
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(L, name), name
    assert L.smmc_abi_version() == _lib.ABI_VERSION == 4
    # and nothing else with the C ABI's prefix leaves the library (helpers shared between its translation
    # units are hidden)
    import subprocess
    from stock_market_monte_carlo_amd import build
    out = subprocess.check_output(["nm", "-D", "--defined-only", build.LIB]).decode()
    exported = {ln.split()[2] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T" and ln.split()[2].startswith("smmc_")}
    assert exported == declared, exported ^ declared


def test_struct_layout_matches_header(L):
    from stock_market_monte_carlo_amd import _lib
    assert C.sizeof(_lib.Sim) == 72 and C.sizeof(_lib.Stats) == 64
    assert L.smmc_stats_bytes(0) == 64 and L.smmc_stats_bytes(100) == 864


def test_host_scalar_functions_match_oracle(L, oracle):
    import stock_market_monte_carlo_amd as S
    rng = np.random.default_rng(3)
    f = rng.uniform(0.5, 1e7, 3000).astype(np.float32)
    r = rng.normal(0.6, 6.0, 3000).astype(np.float32)
    for a, b in zip(f, r):
        assert np.float32(S.update_fund(a, b)).view(np.uint32) == np.float32(oracle.update_fund(a, b)).view(np.uint32)
    rets = rng.normal(0.6, 4.3, 1000).astype(np.float32)
    for p in (0, 1, 360, 1000):
        assert np.array_equal(S.many_updates(1000.0, rets, p).view(np.uint32),
                              oracle.many_updates(1000.0, rets, p).view(np.uint32))
    with pytest.raises(ValueError):
        S.many_updates(1000.0, rets[:5], 6)


def test_stats_merge_host(L):
    from stock_market_monte_carlo_amd import _lib
    from stock_market_monte_carlo_amd.engine import merge_stats_bytes, stats_from_bytes

    def rec(count, below, s, mn, mx, hist):
        h = _lib.Stats(count, below, 1, 2, s, s * s, mn, mx, len(hist), 0)
        return bytes(h) + np.asarray(hist, dtype=np.uint64).tobytes()

    m = stats_from_bytes(merge_stats_bytes([rec(10, 3, 1.5, 2.0, 9.0, [1, 2, 3]), rec(5, 1, 2.5, 1.0, 4.0, [4, 0, 1])]))
    assert (m.count, m.below, m.underflow, m.overflow) == (15, 4, 2, 4)
    assert m.sum == 4.0 and m.min == 1.0 and m.max == 9.0 and m.hist.tolist() == [5, 2, 4]
    with pytest.raises(_lib.SmmcError):
        merge_stats_bytes([rec(1, 0, 1.0, 1.0, 1.0, [1]), rec(1, 0, 1.0, 1.0, 1.0, [1, 2])])


def test_read_historical_returns(tmp_path):
    import stock_market_monte_carlo_amd as S
    t = S.read_historical_returns(os.path.join(ROOT, "data", "SP500_monthly_returns.csv"))
    assert t.shape == (1127,) and t.dtype == np.float32
    p = tmp_path / "x.csv"
    p.write_text("Date,other,returns\n2000-01,1,\n2000-02,2,1.25\n2000-03,3,-0.5\n")
    assert S.read_historical_returns(str(p)).tolist() == [1.25, -0.5]
    p.write_text("Date,ret\n2000-01,1\n")
    with pytest.raises(ValueError):
        S.read_historical_returns(str(p))


def test_no_gpu_means_loud_failure(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import stock_market_monte_carlo_amd as S
    with pytest.raises(S.SmmcError):
        S.Engine(0)
    h = C.c_void_p()
    rc = L.smmc_engine_create(0, None, C.byref(h))
    assert rc == -3 and b"no HIP device" in L.smmc_last_error()


def test_product_does_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/: not the
    package, not the headers, not the development tools."""
    sources = [os.path.join(dirpath, fn) for top in ("stock_market_monte_carlo_amd", "include", "tools")
               for dirpath, _, files in os.walk(os.path.join(ROOT, top)) for fn in files
               if fn.endswith((".py", ".cpp", ".hip", ".h", ".sh"))]
    assert len(sources) > 20
    for path in sources:
        text = open(path).read()
        for needle in ("libsmmc_oracle", "from oracle", "import oracle", "oracle.py", "orc_"):
            assert needle not in text, (path, needle)
        for line in text.splitlines():
            if line.lstrip().startswith("#include"):
                assert "oracle" not in line, (path, line)


def test_struct_sizes_agree_with_the_c_compiler(tmp_path):
    import subprocess
    from stock_market_monte_carlo_amd import _lib
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "smmc.h"\nint main(void){printf("%zu %zu\\n", sizeof(smmc_sim), sizeof(smmc_stats));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    a, b = subprocess.check_output([str(exe)]).split()
    assert int(a) == C.sizeof(_lib.Sim) and int(b) == C.sizeof(_lib.Stats)


def test_every_environment_knob_of_the_library_is_documented():
    """Every SMMC_* variable the product reads (getenv in csrc/) has a row in INTEGRATION.md's table."""
    import re
    names = set()
    csrc = os.path.join(ROOT, "stock_market_monte_carlo_amd", "csrc")
    for dirpath, _, files in os.walk(csrc):
        for fn in files:
            if fn.endswith((".cpp", ".hip", ".h")):
                names |= set(re.findall(r'getenv\("(SMMC_[A-Z0-9_]+)"\)', open(os.path.join(dirpath, fn)).read()))
    assert len(names) >= 15
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
