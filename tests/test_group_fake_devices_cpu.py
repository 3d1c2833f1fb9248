"""smmc_group_* and the C++ drop-in's n_gpus calls with THREE and with EIGHT distinct devices, on the CPU, under ThreadSanitizer
and AddressSanitizer + UBSan (the pool gives one GPU, so the per-device host threads, the
once-only registration of the caller's buffer and the record merge had only ever run with one device or with one
device listed several times).  The devices are tests/cpp/fake_hip.cpp's -- host memory behind the HIP runtime's
own entry points, linked instead of libamdhip64 -- and a "launch" (tests/cpp/launch_fake.cpp) writes a known
function of the global path id, so tests/cpp/group_fake_devices.cpp can check that every id of a sharded, chunked,
multi-threaded run landed in its place exactly once and that the merged record is the record of all of them.
Reference: mc_simulations_multi_gpu_launcher_async, src/simulations.cu:576-655."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "oracle", "_san")


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_san/group_fake_tsan", "_san/group_fake_asan",
                           "_san/fake_rccl_tsan/librccl.so.1", "_san/fake_rccl_asan/librccl.so.1"], stdout=subprocess.DEVNULL)


def _run(exe, **env):
    e = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
             TSAN_OPTIONS="halt_on_error=1", SMMC_VERBOSE="1", FAKE_HIP_DEVICES="3")
    for k in ("SMMC_PIN_HOST", "SMMC_HOST_CHUNK_PATHS", "SMMC_DEVICE_MAP", "SMMC_GROUP_MERGE", "SMMC_SEED", "SMMC_STREAM"):
        e.pop(k, None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([os.path.join(SAN, exe)], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    text = r.stdout + r.stderr
    assert r.returncode == 0 and "group_fake_devices: ok" in r.stdout, text[-3000:]
    assert "Sanitizer" not in text and "runtime error" not in text, text[-3000:]
    # SMMC_VERBOSE reports a registration that fell back to pageable copies: with one owner per page there is none
    assert "hipHostRegister" not in text, text[-3000:]
    return text


@pytest.mark.parametrize("exe", ["group_fake_tsan", "group_fake_asan"])
def test_three_device_group_and_dropin(built, exe):
    text = _run(exe)
    assert "group of 3 device(s)" in text and "shard 2 on device 2" in text


def test_eight_device_group_and_dropin_under_tsan(built):
    """The node the reference and BASELINE configs[3] / [4] are quoted on: eight devices -- eight host threads, eight
    shards with a remainder, one registration of the caller's buffer, the merge of eight records; two groups over the
    same eight devices at once."""
    text = _run("group_fake_tsan", FAKE_HIP_DEVICES=8)
    assert "group of 8 device(s)" in text and "shard 7 on device 7" in text


@pytest.mark.parametrize("exe, devices", [("group_fake_tsan", 3), ("group_fake_asan", 3), ("group_fake_tsan", 8)])
def test_rccl_merge_with_several_devices_over_a_fake_librccl(built, exe, devices):
    """SMMC_MERGE_RCCL with G > 1 -- ncclCommInitAll over G devices, ONE ncclGroupStart / ncclGroupEnd bracket per call
    with two all-reduces per device, the merged integers read back from device 0, doubles and min / max on the host --
    had only ever run with G = 1 (the pool has one GPU).  tests/cpp/fake_rccl.cpp is the librccl.so.1 that
    smmc_group.cpp's dlopen finds here: host memory, completes inside ncclGroupEnd, and REFUSES what real RCCL would
    answer with a hang (a rank missing from the bracket, unequal numbers of collectives) or with garbage (ranks that
    disagree on count / type / operation).  Checked: the merged record equals the host merge's bit for bit (64, 0 and
    1000 buckets) and the definition's; one bracket and 2 (1 without buckets) collectives per call; a collective that
    fails on the second device gives SMMC_ERR_HIP with RCCL's words and the next call works; every communicator is
    destroyed; one device listed twice is refused; the C++ drop-in under SMMC_GROUP_MERGE=rccl.  It cannot show that
    real RCCL over xGMI works -- only that what this library asks of it is well-formed for G ranks."""
    san = "tsan" if exe.endswith("tsan") else "asan"
    text = _run(exe, FAKE_RCCL=1, FAKE_HIP_DEVICES=devices,
                LD_LIBRARY_PATH=os.path.join(SAN, f"fake_rccl_{san}") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    assert f"rccl merge over {devices} fake devices: ok so far" in text and "RCCL communicator in" in text


@pytest.mark.parametrize("env", [dict(SMMC_PIN_HOST="chunk"), dict(SMMC_PIN_HOST="0"), dict(SMMC_HOST_CHUNK_PATHS=65536),
                                 dict(SMMC_PIN_HOST="chunk", SMMC_HOST_CHUNK_PATHS=131072),
                                 # eight devices reporting 17 chunks each: the progress callback must see a total that
                                 # never goes back (it did, 7 then 5, until the total was advanced under the lock)
                                 dict(SMMC_HOST_CHUNK_PATHS=65536, FAKE_HIP_DEVICES=8)])
def test_pinning_policies_and_chunk_lengths_with_three_devices(built, env):
    """Every SMMC_PIN_HOST policy and short chunks (46 per shard; 17 per shard on eight devices): the same checks; under
    ThreadSanitizer."""
    _run("group_fake_tsan", **env)
