"""CPU sanitizer builds (the reference's own race-check recipe is README.md:107-109):
`make -C oracle asan tsan` compiles the oracle, the as-reference C++ leg, the libstdc++ pin program and
the product's HOST-side code (csrc/smmc_capi.cpp + csrc/smmc_dropin.cpp against launch stubs) with
ASan + UBSan, and the product driver again with TSan; every binary must run clean.  CPU only: GPU
sanitizer runs are not available on this pool."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan", "tsan"], stdout=subprocess.DEVNULL)


def _run(exe, *args, env=None):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
             TSAN_OPTIONS="halt_on_error=1")
    e.update(env or {})
    return subprocess.run([os.path.join(SAN, exe), *args], cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)


def _clean(r):
    text = r.stdout + r.stderr
    return r.returncode == 0 and "Sanitizer" not in text and "runtime error" not in text, text[-3000:]


def test_oracle_under_asan_ubsan(built):
    r = _run("oracle_asan")
    ok, text = _clean(r)
    assert ok and "oracle_driver: ok" in r.stdout, text


def test_as_reference_leg_under_asan_ubsan(built):
    r = _run("asref_asan")
    ok, text = _clean(r)
    assert ok and "asref_driver: ok" in r.stdout, text


def test_pin_program_under_asan_ubsan(built, table, tmp_path):
    path = tmp_path / "table.txt"
    np.savetxt(path, table, fmt="%.9g")
    r = _run("pin_asan", str(path))
    ok, text = _clean(r)
    assert ok and '"mt19937_default_10000th": 4123659995' in r.stdout, text


def test_product_host_code_under_asan_ubsan(built):
    r = _run("host_asan")
    ok, text = _clean(r)
    assert ok and "host_sanitize: ok" in r.stdout, text


def test_product_host_code_under_tsan(built):
    r = _run("host_tsan")
    ok, text = _clean(r)
    assert ok and "host_sanitize: ok" in r.stdout, text
