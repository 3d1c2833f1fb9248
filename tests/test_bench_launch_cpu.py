"""bench.py's own rank launcher, on the CPU: `python bench.py --gpus N` must start N ranks, have
them rendezvous, run the statistics gather and print exactly ONE JSON line (the driver calls it that
way: no torchrun).  `--launch-check` runs the control path without a GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "SMMC_BENCH_CHILD"):
        e.pop(k, None)
    return e


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("n", [2, 3, 8])  # 8 = the rank count of the driver's SCALE run
def test_plain_invocation_launches_its_own_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--backend", "gloo", "--launch-check"], cwd=ROOT,
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == n and d["ranks"] == n and d["backend"] == "gloo" and d["launcher"] == "self"
    assert len(d["devices"]) == n and all(f"rank {i}:" in d["devices"][i] for i in range(n))


def test_ranks_made_by_torchrun_are_used_as_they_are():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--launch-check"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["ranks"] == 2 and lines[0]["launcher"] == "external"


def test_launcher_rank_count_mismatch_is_an_error():
    e = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_a_failing_rank_fails_the_run_and_leaves_no_process_behind():
    # without a GPU every real (non launch-check) rank exits with an error: the parent must report it
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=dict(_env(), HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not _json_lines(r.stdout)


def test_presets_name_the_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert sorted(bench.CONFIGS) == list(range(len(base)))
    assert bench.CONFIGS[1]["paths_per_gpu"] == 100_000_000 and bench.CONFIGS[1]["periods"] == 360
    assert bench.CONFIGS[2]["mode"] == "table"
    assert bench.CONFIGS[3]["total_paths"] == 10 ** 9 and bench.CONFIGS[3]["outputs"] == "stats"
    assert bench.CONFIGS[4]["total_paths"] == 10 ** 9 and bench.CONFIGS[4]["periods"] == 1000
    assert bench.CONFIGS[4]["outputs"] == "host"
    for k, c in bench.CONFIGS.items():
        assert f"configs[{k}]" in c["name"]


def test_a_rank_that_dies_mid_run_ends_the_run_within_the_grace_period(tmp_path):
    """One rank exits after the rendezvous, the other never returns from its step (a rank blocked in a
    collective with a dead peer): the parent waits its grace period (20 s by default, 3 s here), stops
    exactly the ranks it started, and exits with the dead rank's code -- bench.py's `failed_at` branch."""
    import time
    pid_file = str(tmp_path / "rank_pid")
    env = dict(_env(), SMMC_BENCH_TEST_DIE_RANK="1", SMMC_BENCH_TEST_PID_FILE=pid_file, SMMC_BENCH_GRACE="3")
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--launch-check"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert r.returncode == 7, (r.returncode, r.stderr[-1000:])
    assert not _json_lines(r.stdout)
    assert took < 120  # start-up of two ranks + 3 s grace + the stop; never the 600 s the survivor would sleep
    survivor = int(open(pid_file + ".0").read())
    time.sleep(0.5)
    alive = True
    try:
        os.kill(survivor, 0)
    except ProcessLookupError:
        alive = False
    except PermissionError:
        pass
    assert not alive, "the surviving rank was left behind"
