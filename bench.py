#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Monte-Carlo returns engine.

Metric (BASELINE.json): simulated paths/sec at 360 periods.  One "step" = one pass of
the hot path over one batch of synthetic input.  `--config K` selects BASELINE.json's
`configs[K]`; the default (K = 1) is the configuration the metric is quoted on:
Gaussian returns, 360 periods x 1e8 paths per GPU, producing the final value of every
path in HBM, the per-256-path block means/variances and the fused statistics record
(sum, sum of squares, below-count, 100-bucket histogram).

  --config 0  configs[0]  360 x 1e6 paths, Gaussian (the reference's CPU-runnable size)
  --config 1  configs[1]  360 x 1e8 paths per GPU, Gaussian, final values + block means + statistics  [default]
  --config 2  configs[2]  360 x 1e8 paths per GPU, historical table staged in LDS
  --config 3  configs[3]  360 x 1e9 paths IN TOTAL, sharded over the ranks, statistics only,
                          one RCCL all_gather of the ~900-byte record per step
  --config 4  configs[4]  1000 x 1e9 paths IN TOTAL, sharded; final values into pinned HOST memory
                          through the chunked kernel / side-stream D2H pipeline (PCIe inside the
                          timed region: that IS this configuration's job)

Multi-GPU: one process per GPU.  `python bench.py --gpus N` starts its N ranks itself
(the parent never touches the GPU; it relays rank 0's JSON line and exits with the
children's return code); under `python -m torch.distributed.run ... bench.py --gpus N`
the ranks torchrun made are used as they are.  Paths shard by contiguous global id ranges
(reference: n_gpus is just an argument, examples/benchmark_mc_gpu.cpp:52-69, split at
src/simulations.cu:599-607); the only exchange is the statistics record, merged in rank order.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      HBM roofline of the dominant kernel (paths_kernel): algorithmic bytes
                (4 B per path, the coalesced final-value store) / its HIP-event time.
                The kernel is VALU-bound by design, so this fraction is tiny; the
                `valu` object carries the roof that actually binds.
  cpu_baseline  the reference's CPU algorithm (oracle engine R: mt19937 + Lemire +
                update_fund, OpenMP hw-1 threads) timed on this host on a bounded sample,
                with deterministic seeds (`value`) and as the reference really does it,
                one std::random_device per path (`as_reference_value`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_PERIODS = 360
PATHS_PER_GPU = 100_000_000
SEED = 0x5EED5EED5EED5EED
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9  # CUs x SIMDs x lanes/clk x max clock = 7.86e13 lane-ops/s
# VALU instructions per path-period of paths_kernel's inner loop, read off the gfx950 ISA by
# tools/isa_loop_count.py (tests/test_measurement_cpu.py asserts these equal the built library's).
VALU_INSTS_PER_STEP = {"gaussian": 70 / 4, "table": 84 / 8}  # Gaussian: counter stream v3 (v2: 122 / 4)
VALU_CHECK_PER_STEP = {"gaussian": 2 / 4, "table": 2 / 8}  # the range-checked divide: two compares per Philox block
# --stream ref (the reference CPU engine's per-path mt19937 stream, ref_windowed_kernel): VALU instructions per
# step of the 397-step seed run-up, per output below output 227 and per output from 227 on (same ISA test)
REF_VALU = {"runup_step": 24 / 8, "output_lo": 100 / 4, "output_hi": 128 / 4}
REF_WINDOW = (397, 227, 454)  # run-up steps, first output of the second stretch, longest path of the windowed kernel
# ref_tree_kernel (paths of 455 .. 1816 periods): first output of each stretch and the VALU instructions of its
# written-out 4-output loop (tests/test_measurement_cpu.py re-derives them from the built kernel)
REF_TREE_STRETCHES = ((0, 100 / 4), (227, 128 / 4), (454, 160 / 4), (623, 180 / 4), (681, 208 / 4), (850, 228 / 4), (908, 260 / 4),
                      (1077, 280 / 4), (1135, 308 / 4), (1246, 328 / 4), (1304, 348 / 4), (1362, 380 / 4), (1473, 400 / 4),
                      (1531, 420 / 4), (1589, 448 / 4), (1700, 468 / 4), (1758, 488 / 4))
REF_TREE_MAX = 1816
REF_CHECK_VALU = 4  # checked divide (kDivChecked): every 8 periods two float compares, a select and the flag's v_cmp


def ref_valu_per_path(periods):
    """VALU instructions the state-free reference-stream kernels spend on one path: ref_windowed_kernel
    (periods <= 454) or ref_tree_kernel (<= 1816), the same counts where both apply."""
    total = REF_WINDOW[0] * REF_VALU["runup_step"]
    starts = [t for t, _ in REF_TREE_STRETCHES] + [REF_TREE_MAX]
    for (first, per_output), nxt in zip(REF_TREE_STRETCHES, starts[1:]):
        total += max(min(periods, nxt) - first, 0) * per_output
    return total
# HBM bytes per launch measured by the round's rocprofv3 PMC passes (tools/pmc_traffic.py writes it)
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")

CONFIGS = {
    0: dict(mode="gaussian", periods=360, paths_per_gpu=1_000_000, total_paths=None, outputs="all",
            name="BASELINE configs[0] size on the GPU: Gaussian returns, 360 periods x 1e6 paths"),
    1: dict(mode="gaussian", periods=360, paths_per_gpu=PATHS_PER_GPU, total_paths=None, outputs="all",
            name="BASELINE configs[1]: Gaussian returns, 360 periods x 1e8 paths, 1xMI355X, final-value only + "
                 "block-reduce mean"),
    2: dict(mode="table", periods=360, paths_per_gpu=PATHS_PER_GPU, total_paths=None, outputs="all",
            name="BASELINE configs[2]: historical S&P500 monthly returns (LDS-staged table), 360 periods x 1e8 paths, "
                 "1xMI355X"),
    3: dict(mode="gaussian", periods=360, paths_per_gpu=None, total_paths=1_000_000_000, outputs="stats",
            name="BASELINE configs[3]: Gaussian returns, 360 periods x 1e9 paths sharded across the GPUs, RCCL "
                 "gather of the statistics record (histogram)"),
    4: dict(mode="gaussian", periods=1000, paths_per_gpu=None, total_paths=1_000_000_000, outputs="host",
            name="BASELINE configs[4]: Gaussian returns, 1000 periods x 1e9 paths sharded across the GPUs, final "
                 "values to pinned host memory, D2H overlapped on a side HIP stream"),
}


def load_table():
    vals = []
    with open(os.path.join(ROOT, "data", "SP500_monthly_returns.csv")) as f:
        col = f.readline().strip().split(",").index("returns")
        for line in f:
            cell = line.rstrip("\n").split(",")[col]
            if cell:
                vals.append(np.float32(cell))
    return np.array(vals, dtype=np.float32)


def usable_cores():
    """CPU cores this process can really use: affinity mask capped by the cgroup CPU quota."""
    if os.environ.get("SMMC_CPU_CORES"):
        return max(1, int(os.environ["SMMC_CPU_CORES"]))
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 2
    quota = None
    try:  # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        cores = min(cores, max(1, int(quota + 0.5)))
    return cores


def cpu_baseline(table, budget_s=12.0, n_periods=N_PERIODS):
    """Times the oracle's reference-faithful engine (R) -- the checker, used here only as
    the CPU baseline the metric asks for."""
    from oracle import oracle as O
    O.build()
    # the reference uses hardware_concurrency() - 1 threads (src/simulations.cpp:218-219);
    # count the cores this process may actually run on, not the host's
    threads = max(1, usable_cores() - 1)
    n, dt, used = 50_000, 0.0, threads
    while True:  # grow the sample until it is >= 2/3 of the budget of CPU wall time
        t0 = time.perf_counter()
        _, used = O.ref_mc_simulations(n, n_periods, 1000.0, table, 12345, n_threads=threads)
        dt = time.perf_counter() - t0
        if dt >= budget_s * 2 / 3 or n >= 200_000_000:
            break
        n = int(n * min(max(budget_s / max(dt, 1e-3), 1.5), 20.0))
        n -= n % 1000
    # the same engine on ONE thread (SURVEY 8d asks for both figures): ~3 s
    n1 = max(1000, int(n / dt * 3.0 / max(used, 1)))
    n1 -= n1 % 1000
    t0 = time.perf_counter()
    O.ref_mc_simulations(n1, n_periods, 1000.0, table, 12345, n_threads=1)
    dt1 = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "paths/s", "cores": used, "kind": "port",
           "single_thread_value": n1 / dt1,
           "sample": f"{n} paths x {n_periods} periods, table mode (T={table.size}), oracle engine R "
                     f"(per-path mt19937 + Lemire + update_fund, OpenMP dynamic blocks of 1000, deterministic "
                     f"seeds: no per-path std::random_device), {dt:.1f} s; single thread: {n1} paths, {dt1:.1f} s"}
    # BASELINE configs[0] as written -- "Gaussian returns, single-thread CPU reference (fixed seed)": the
    # reference's Gaussian demo path (std::default_random_engine + std::normal_distribution<float>(0.5, 0.83333),
    # src/simulations.cpp:41-67) beside the GPU's Gaussian headline; one thread, then all of them
    try:
        ng, dtg = 100_000, 0.0
        while True:
            t0 = time.perf_counter()
            O.asref_gaussian_mc(ng, n_periods, 1000.0, 0.5, 0.83333, 12345, n_threads=1)
            dtg = time.perf_counter() - t0
            if dtg >= 2.0 or ng >= 50_000_000:
                break
            ng = int(ng * min(max(3.0 / max(dtg, 1e-3), 1.5), 20.0))
            ng -= ng % 1000
        nga = max(1000, int(ng / dtg * 3.0 * used))
        nga -= nga % 1000
        t0 = time.perf_counter()
        O.asref_gaussian_mc(nga, n_periods, 1000.0, 0.5, 0.83333, 12345, n_threads=threads)
        dtga = time.perf_counter() - t0
        out["gaussian_single_thread_value"] = ng / dtg
        out["gaussian_value"] = nga / dtga
        out["gaussian_sample"] = (f"{ng} paths x {n_periods} periods on 1 thread, {dtg:.1f} s; {nga} paths on {used} threads, "
                                  f"{dtga:.1f} s: std::default_random_engine + std::normal_distribution<float>(0.5, 0.83333) "
                                  f"per path, fixed seed, update_fund (oracle/asref_cpu.cpp: BASELINE configs[0] as written)")
    except Exception as ex:
        out["gaussian_single_thread_value"] = out["gaussian_value"] = None
        out["gaussian_sample"] = f"unavailable: {ex}"
    # variant (i) of SURVEY 8d: as the reference does it, a fresh std::random_device seeding a fresh
    # mt19937 for EVERY path (src/simulations.cpp:245-247) -- set-up dominated and OS dependent
    try:
        nr = 20_000  # grown until it runs >= 3 s: the per-path cost is OS dependent (4-45 us measured)
        while True:
            t0 = time.perf_counter()
            O.asref_mc_simulations(nr, n_periods, 1000.0, table, n_threads=threads)
            dtr = time.perf_counter() - t0
            if dtr >= 3.0 or nr >= n:
                break
            nr = min(n, int(nr * min(max(4.0 / max(dtr, 1e-3), 1.5), 10.0)))
            nr -= nr % 1000
        out["as_reference_value"] = nr / dtr
        out["as_reference_sample"] = (f"{nr} paths, one std::random_device + mt19937 seeding per path as "
                                      f"src/simulations.cpp:245-247, {used} threads, {dtr:.1f} s")
    except Exception as ex:  # the C++ leg is optional: never lose the line to it
        out["as_reference_value"] = None
        out["as_reference_sample"] = f"unavailable: {ex}"
    return out


def hbm_bound_kernels(eng, S, final, mode):
    """The path's HBM-bound neighbours, timed live with HIP events on the
    engine's stream AFTER the timed region: statistics and exact quartiles of the resident final
    values (4 B read per value per pass), and keepdata (4 (P+1) B written per path)."""
    out = {}
    n = final.numel()

    def timed(fn, reps):
        fn()
        eng.sync()
        eng.timing(True)
        for _ in range(reps):
            fn()
        ms, k = eng.kernel_ms()
        eng.timing(False)
        return ms / max(k, 1)

    ms = timed(lambda: eng.values_stats(final, 1000.0, 100, 0.0, 20000.0), 10)
    out["values_stats"] = {"bytes_per_launch": 4.0 * n, "kernel_ms": ms, "GBps": 4.0 * n / ms / 1e6,
                           "frac_of_peak": 4.0 * n / ms / 1e6 / HBM_PEAK_GBS}
    ms = timed(lambda: eng.quartiles(final), 5)  # 3 histogram passes per call, each timed
    out["quartiles_radix_pass"] = {"bytes_per_launch": 4.0 * n, "kernel_ms": ms, "GBps": 4.0 * n / ms / 1e6,
                                   "frac_of_peak": 4.0 * n / ms / 1e6 / HBM_PEAK_GBS}
    nk, p = 4_000_000, N_PERIODS
    sim = S.Engine.make_sim(nk, p, mode, SEED)
    traj, _ = eng.simulate_keepdata(sim, want_final=False)
    ms = timed(lambda: _lib_keepdata(eng, sim, traj), 20)
    b = 4.0 * nk * (p + 1)
    out["keepdata"] = {"bytes_per_launch": b, "kernel_ms": ms, "GBps": b / ms / 1e6, "frac_of_peak": b / ms / 1e6 / HBM_PEAK_GBS,
                       "n_paths": nk}
    # a yardstick from the same box and the same minute (boxes differ by +- 5 % in what their HBM streams): ATen's own
    # fill of the trajectory buffer and its sum over the final values -- the same bytes, nothing computed
    try:
        import torch

        def torch_timed(fn, reps):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps

        eng.sync()
        fill_ms = torch_timed(lambda: traj.fill_(1.0), 10)
        sum_ms = torch_timed(lambda: final.sum(), 20)
        out["box_yardstick"] = {"fill_GBps": b / fill_ms / 1e6, "fill_bytes": b, "sum_GBps": 4.0 * n / sum_ms / 1e6, "sum_bytes": 4.0 * n,
                                "what": "torch.Tensor.fill_ over the keepdata buffer and torch.sum over the final values on this box, "
                                        "torch events on torch's stream: a vendor kernel moving the same bytes, not a ceiling"}
        out["keepdata"]["vs_box_fill"] = out["keepdata"]["GBps"] / out["box_yardstick"]["fill_GBps"]
        out["values_stats"]["vs_box_sum"] = out["values_stats"]["GBps"] / out["box_yardstick"]["sum_GBps"]
    except Exception as ex:  # a yardstick, never worth the line
        out["box_yardstick"] = {"error": str(ex)}
    # LAST (these loops leave the device half idle, and the clocks of the measurements after them would show it):
    # the whole quartiles call as a caller sees it (update_quartiles, examples/visualize_returns_cpu_v2.cpp:83-111): the ranks go up,
    # three passes and their picks run, five values come back; and the same on 1e6 values, where the launches ARE the call
    t0 = time.perf_counter()
    for _ in range(10):
        eng.quartiles(final)
    out["quartiles_radix_pass"]["call_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    small = final[:1_000_000]
    eng.quartiles(small)
    t0 = time.perf_counter()
    for _ in range(50):
        eng.quartiles(small)
    out["quartiles_radix_pass"]["call_ms_1e6_values"] = (time.perf_counter() - t0) / 50 * 1e3
    return out


def _lib_keepdata(eng, sim, traj):
    import ctypes as C
    from stock_market_monte_carlo_amd import _lib
    _lib.check(eng._L.smmc_engine_simulate_keepdata(eng._h, C.byref(sim), C.c_void_p(traj.data_ptr()), None))


def pmc_traffic(mode, n, periods, outputs, want="bytes"):
    """HBM bytes per paths_kernel launch from the rocprofv3 PMC passes committed under profiles/
    (bench.py cannot run rocprofv3 on itself): (bytes, source) or (None, None) when this workload
    was not profiled.  want="valu": the entry's priced period loop (tools/valu_model.py) instead of the bytes."""
    try:
        table = json.load(open(PMC_TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, None
    rec = table.get(f"{mode}|{n}|{periods}|{outputs}")
    if not rec and want == "valu":  # the loop's price does not depend on the launch's size: any entry of the same kernel
        rec = next((r for k, r in sorted(table.items()) if k.split("|")[0] == mode and int(k.split("|")[2]) == periods and r.get("valu")), None)
    if not rec:
        return None, None
    # the figure belongs to the build that was profiled: kernel sources + compiler flags must be the ones
    # in this tree (the compiled ISA is compared by tests/test_measurement_cpu.py)
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_loop_count as I
        if rec.get("source_sha256") != I.source_digest(I.traffic_kernel_of(f"{mode}|{n}|{periods}|{outputs}")):
            return None, f"stale: the kernel sources changed since {rec.get('source')}"
    except Exception as ex:
        return None, f"unverified: {ex}"
    if want == "valu":
        return rec.get("valu"), (rec.get("valu") or {}).get("weights_source")
    return float(rec["bytes"]), rec.get("source")


def _build_digest():
    """sha256 of the sources + flags libsmmc_hip.so was built from (smmc_build_digest; the loader has already
    checked it against the tree's sources)."""
    from stock_market_monte_carlo_amd import _lib
    return _lib.build_digest()


# ---- launching the ranks ---------------------------------------------------------------------


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes, relay rank 0's
    stdout (the JSON line), return the worst return code.  This parent never initialises the GPU (it
    has not even imported torch) and nothing here execs: children are ordinary subprocesses."""
    port = _free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SMMC_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    rc = 0
    out0 = b""
    try:
        deadline = time.time() + float(os.environ.get("SMMC_BENCH_TIMEOUT", "1500"))
        grace = float(os.environ.get("SMMC_BENCH_GRACE", "20"))  # how long the other ranks get once one has failed
        pending = set(range(n_ranks))
        failed_at = None
        while pending:
            for r in sorted(pending):
                if r == 0:
                    try:
                        o, _ = procs[0].communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    failed_at = failed_at or time.time()
            if pending and ((failed_at and time.time() - failed_at > grace) or time.time() > deadline):
                rc = rc or 124  # a rank died or the run overran: the others would wait for it forever
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # exactly the processes started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    # ONE JSON line on stdout; whatever else rank 0 printed there (gloo's connection banner) goes to stderr
    lines = out0.decode(errors="replace").splitlines()
    for i, line in enumerate(lines):
        if rc == 0 and line.startswith('{"metric"') and "--launch-check" not in argv and not os.environ.get("SMMC_BENCH_NO_GROUP"):
            # the ranks are gone: the SAME workload once more through the one-process C-ABI launcher
            # (smmc_group_*, RCCL all-reduce and host merge), from a fresh child of this GPU-less parent
            try:
                d = json.loads(line)
                d["group_single_process"] = run_group_child(n_ranks, argv)
                lines[i] = json.dumps(d)
            except ValueError:
                pass
    for line in lines:
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    return rc


def group_child(args):
    """`bench.py --group-child N ...`: the same workload through the C ABI's ONE-process launcher,
    smmc_group_* (csrc/smmc_group.cpp; reference: mc_simulations_multi_gpu_launcher_async behind
    mc_simulations_gpu(n_gpus), src/simulations.cu:576-680) -- one host thread and one engine per device, the
    per-device records merged by ONE grouped ncclAllReduce (merge="rccl") and by the host loop (merge="host").
    A fresh process started by the parent of the ranks (or by rank 0 after its process group is gone); prints
    one JSON object.  Statistics-only for configs 0-3 (the group entry has no device-resident output), final
    values into pinned host memory for config 4."""
    import torch
    import stock_market_monte_carlo_amd as S
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        print(json.dumps({"error": "no device visible"}))
        return
    G = args.group_child
    preset = CONFIGS[args.config]
    mode_name = args.mode or preset["mode"]
    periods = args.periods if args.periods is not None else preset["periods"]
    outputs = args.outputs or preset["outputs"]
    total_paths, per_gpu = preset["total_paths"], preset["paths_per_gpu"]
    if args.total_paths is not None:
        total_paths, per_gpu = args.total_paths, None
    elif args.paths_per_gpu is not None:
        total_paths, per_gpu = None, args.paths_per_gpu
    n_all = total_paths if total_paths is not None else per_gpu * G
    to_host = outputs == "host"
    mode = S.MODE_GAUSSIAN if mode_name == "gaussian" else S.MODE_TABLE
    sim = S.Engine.make_sim(n_all, periods, mode, SEED, initial_capital=1000.0, gauss_mean=0.5, gauss_std=0.83333,
                            n_bins=100, hist_lo=0.0, hist_hi=20000.0)
    host = torch.empty(n_all, dtype=torch.float32, pin_memory=True).numpy() if to_host else None
    out = {"devices": G, "visible_devices": n_dev, "paths": n_all, "n_periods": periods, "mode": mode_name,
           "outputs": "host" if to_host else "stats", "steps": args.steps, "warmup": args.warmup,
           "entry": "smmc_group_simulate (one process, one host thread + engine per device)"}
    records = {}
    for merge in ("rccl", "host"):
        leg = {}
        try:
            if merge == "rccl" and n_dev < G:
                raise RuntimeError(f"needs {G} distinct devices, {n_dev} visible")
            devices = list(range(G)) if n_dev >= G else [i % n_dev for i in range(G)]
            t0 = time.perf_counter()
            grp = S.Group(devices, merge=merge)
            leg["create_ms"] = (time.perf_counter() - t0) * 1e3
            grp.set_table(load_table())
            for _ in range(args.warmup):
                grp.simulate(sim, out=host, want_final=to_host, want_stats=True)
            merges = []
            t0 = time.perf_counter()
            for _ in range(args.steps):
                _, st, _ = grp.simulate(sim, out=host, want_final=to_host, want_stats=True)
                merges.append(grp.timings()[2])
            dt = time.perf_counter() - t0
            engines_ms, comm_ms, _ = grp.timings()
            grp.close()
            assert st.count == n_all, (st.count, n_all)
            records[merge] = (st.count, st.below, st.underflow, st.overflow, st.min, st.max, st.hist.tobytes(), st.sum, st.sumsq)
            leg.update({"value": n_all * args.steps / dt, "unit": "paths/s", "ms_per_step": dt / args.steps * 1e3,
                        "merge_ms": sum(merges) / len(merges), "engines_ms": engines_ms, "comm_init_ms": comm_ms,
                        "device_list": devices, "result": {"mean": st.mean, "std": st.std, "below_initial": st.below}})
        except Exception as ex:  # one leg failing must not lose the other
            leg["error"] = f"{type(ex).__name__}: {ex}"
        out[merge] = leg
    if len(records) == 2:
        out["rccl_equals_host_merge"] = records["rccl"] == records["host"]
    print(json.dumps(out), flush=True)


def run_group_child(n_gpus, argv, timeout_s=None):
    """Starts `bench.py --group-child N <the run's own workload flags>` as a fresh child process (nothing here
    execs), returns its JSON object or {"error": ...}.  Never raises: the headline line must not depend on it."""
    timeout_s = timeout_s or float(os.environ.get("SMMC_BENCH_GROUP_TIMEOUT", "180"))
    keep, skip = [], 0
    for i, a in enumerate(argv):  # drop the launcher's own flags, keep the workload's
        if skip:
            skip -= 1
            continue
        if a in ("--gpus", "--backend", "--hash-shards"):
            skip = 1
            continue
        if a.startswith(("--gpus=", "--backend=", "--hash-shards=")) or a in ("--rehearse-rccl", "--no-cpu-baseline"):
            continue
        keep.append(a)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE",
                                                             "MASTER_ADDR", "MASTER_PORT", "SMMC_BENCH_CHILD",
                                                             "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--group-child", str(n_gpus)] + keep, env=env,
                           capture_output=True, text=True, timeout=timeout_s)
        for line in reversed(p.stdout.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": f"no result (rc {p.returncode}): {p.stderr.strip()[-400:]}"}
    except subprocess.TimeoutExpired:
        return {"error": f"timed out after {timeout_s:.0f} s"}
    except Exception as ex:
        return {"error": f"{type(ex).__name__}: {ex}"}


def launch_check(args, world, rank):
    """CPU rehearsal of the N > 1 control path (no GPU, no kernels): the ranks rendezvous, each builds
    the statistics record of its shard of `--total-paths` synthetic paths, the product's one
    all_gather + rank-order merge runs, and rank 0 prints the JSON line."""
    import torch
    import torch.distributed as dist
    from stock_market_monte_carlo_amd import _lib
    from stock_market_monte_carlo_amd.dist import all_gather_merge_stats, shard_range
    if world > 1:
        dist.init_process_group("gloo")
    # test hook (tests/test_bench_launch_cpu.py): rank SMMC_BENCH_TEST_DIE_RANK dies after the rendezvous and
    # the others never come back from their step (as ranks blocked in a collective with a dead peer can):
    # the parent must give up after its grace period, stop them and exit non-zero
    die = os.environ.get("SMMC_BENCH_TEST_DIE_RANK")
    if die is not None and world > 1:
        if rank == int(die):
            os._exit(7)
        pid_file = os.environ.get("SMMC_BENCH_TEST_PID_FILE")
        if pid_file:
            with open(f"{pid_file}.{rank}", "w") as fh:
                fh.write(str(os.getpid()))
        time.sleep(600)
    n_total = args.total_paths or 10007 * world + 3
    first, count = shard_range(n_total, world, rank)
    n_bins = 16
    hdr = _lib.Stats(count, count // 3, 0, 0, float(count), float(count), 1.0, 1.0, n_bins, 0)
    hist = np.zeros(n_bins, dtype=np.uint64)
    hist[rank % n_bins] = count
    rec = torch.frombuffer(bytearray(bytes(hdr) + hist.tobytes()), dtype=torch.uint8)
    t0 = time.perf_counter()
    merged = all_gather_merge_stats(rec) if world > 1 else None
    dt = time.perf_counter() - t0
    names = [None] * world
    if world > 1:
        dist.all_gather_object(names, f"rank {rank}: cpu (launch check)")
        assert merged.count == n_total and int(merged.hist.sum()) == n_total, (merged.count, n_total)
    else:
        names = ["rank 0: cpu (launch check)"]
    if rank == 0:
        print(json.dumps({"metric": "launch check (no simulation)", "value": 0.0, "unit": "paths/s", "n_gpus": world,
                          "ranks": dist.get_world_size() if world > 1 else 1, "backend": "gloo" if world > 1 else "none",
                          "devices": names, "steps": 0, "warmup": 0, "ms_per_step": dt * 1e3,
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
                          "data": "synthetic", "config": {"workload": "launch check", "total_paths": n_total},
                          "launcher": "self" if os.environ.get("SMMC_BENCH_CHILD") else "external"}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=1,
                    help="BASELINE.json configs[K]; explicit flags below override the preset")
    ap.add_argument("--mode", choices=["gaussian", "table"], default=None)
    ap.add_argument("--paths-per-gpu", type=int, default=None, help="weak scaling: this many paths on every rank")
    ap.add_argument("--total-paths", type=int, default=None, help="strong scaling: this many paths sharded over the ranks")
    ap.add_argument("--periods", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL (the real multi-GPU run); gloo lets several ranks rehearse on ONE GPU")
    ap.add_argument("--outputs", choices=["all", "final", "stats", "host"], default=None,
                    help="all = final values + block means + statistics (configs[1]); final = final values only; "
                         "stats = statistics only (no per-path HBM write); host = final values into pinned host "
                         "memory through the chunked side-stream pipeline (configs[4])")
    ap.add_argument("--stream", choices=["3", "2", "ref"], default="3",
                    help="3 / 2: the build's Philox counter streams (3 is the default and the headline); ref: the "
                         "reference CPU engine's own stream on the device -- per-path mt19937 + libstdc++ Lemire map "
                         "(src/simulations.cpp:240-252), table mode")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="with --gpus 1: still initialise the process group (one rank) and run every collective "
                         "of the N > 1 path -- barrier, the record all_gather, the max-reduce of the time -- so that "
                         "a one-GPU box exercises torch's nccl backend (RCCL)")
    ap.add_argument("--hash-shards", type=int, default=0, metavar="K",
                    help="outputs=host only, after the timed region: report a 64-bit digest of the final values in the "
                         "host buffer -- N > 1: every rank's own buffer, in rank order; N = 1: of the K contiguous shares "
                         "a K-rank run would hold -- so that a multi-rank run can be compared with a one-rank run bit for bit")
    ap.add_argument("--group-child", type=int, default=0, metavar="N",
                    help="internal: run the workload through ONE process's smmc_group over N devices (RCCL all-reduce and "
                         "host merge) and print one JSON object; started by `--gpus N` after its ranks have finished")
    ap.add_argument("--launch-check", action="store_true",
                    help="CPU-only rehearsal of rank launch + rendezvous + the statistics gather (no GPU work)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.group_child:
        return group_child(args)

    have_ranks = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not have_ranks:
        # no launcher made our ranks: make them (BEFORE anything in this process touches torch or HIP)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1")) if have_ranks else 1
    rank = int(os.environ.get("RANK", "0")) if have_ranks else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if have_ranks else 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher made WORLD_SIZE={world} ranks")
    if args.launch_check:
        return launch_check(args, world, rank)

    preset = CONFIGS[args.config]
    mode_name = args.mode or ("table" if args.stream == "ref" else preset["mode"])
    if args.stream == "ref" and mode_name != "table":
        raise SystemExit("--stream ref is the reference's table-draw engine: --mode table")
    periods = args.periods if args.periods is not None else preset["periods"]
    outputs = args.outputs or preset["outputs"]
    total_paths, per_gpu = preset["total_paths"], preset["paths_per_gpu"]
    if args.total_paths is not None:
        total_paths, per_gpu = args.total_paths, None
    elif args.paths_per_gpu is not None:
        total_paths, per_gpu = None, args.paths_per_gpu
    is_preset = (args.mode is None and args.periods is None and args.outputs is None and args.total_paths is None
                 and args.paths_per_gpu is None and args.stream == "3")

    import torch
    import torch.distributed as dist
    import stock_market_monte_carlo_amd as S
    from stock_market_monte_carlo_amd.dist import gather_stats_tensor, records_from_tensor, shard_range
    from stock_market_monte_carlo_amd.engine import merge_stats_bytes, stats_from_bytes

    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("no MI355X visible: bench.py has no CPU path (use --launch-check for the CPU rehearsal)")
    if args.backend == "nccl" and world > n_dev:
        raise SystemExit(f"--gpus {world} with the RCCL backend needs {world} devices, {n_dev} visible "
                         "(use --backend gloo to rehearse several ranks on one GPU)")
    device = local_rank % n_dev
    torch.cuda.set_device(device)
    use_dist = world > 1 or args.rehearse_rccl
    if use_dist:
        if world == 1 and not have_ranks:
            for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"),
                             ("MASTER_PORT", str(_free_port()))):
                os.environ.setdefault(key, val)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    table = load_table()
    eng = S.Engine(device)
    eng.set_table(table)
    mode = S.MODE_GAUSSIAN if mode_name == "gaussian" else S.MODE_TABLE
    if total_paths is not None:  # strong scaling: contiguous id ranges, the remainder kept
        first, n = shard_range(total_paths, world, rank)
        n_all = total_paths
    else:                        # weak scaling
        first, n = rank * per_gpu, per_gpu
        n_all = per_gpu * world
    sim = S.Engine.make_sim(n, periods, mode, SEED, first_path=first, initial_capital=1000.0,
                            gauss_mean=0.5, gauss_std=0.83333, n_bins=100, hist_lo=0.0, hist_hi=20000.0,
                            stream="ref" if args.stream == "ref" else int(args.stream))
    to_host = outputs == "host"
    want_final = outputs in ("all", "final")
    want_chunks = outputs == "all"
    want_stats = outputs in ("all", "stats")
    final = torch.empty(n, dtype=torch.float32, device=eng.tdevice) if want_final else None
    host_final = None
    if to_host:
        # the caller's result buffer, pinned once outside the timed region as the reference does with
        # cudaMallocHost (src/simulations.cu:591-592)
        host_final = torch.empty(n, dtype=torch.float32, pin_memory=True).numpy()
    records = None

    def step():
        # the whole per-step job: simulate this rank's shard, then (N > 1) the one RCCL
        # all_gather of the statistics records, merged in rank order on the host
        nonlocal records
        if to_host:
            eng.simulate_to_host(sim, out=host_final)
            return None
        r = eng.simulate(sim, want_final=want_final, want_chunk_stats=want_chunks, want_stats=want_stats, out=final)
        if want_stats:  # the gathered records stay where the collective left them: no host sync inside the step
            records = gather_stats_tensor(r.stats_raw, force=use_dist) if use_dist else None
        return r

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.timing(True)
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = eng.kernel_ms()
    held_clock_ghz = eng.kernel_clock_ghz()  # sampled by the timed launches' own workgroups (0.0: a kernel without the probe)
    eng.timing(False)

    devices = [None] * world
    mine = f"rank {rank}: cuda:{device} {torch.cuda.get_device_name(device)}"
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.tdevice if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.all_gather_object(devices, mine)
    else:
        devices = [mine]

    host_digests = None
    if to_host and args.hash_shards:
        import hashlib

        def digest(a):
            return hashlib.blake2b(np.ascontiguousarray(a).view(np.uint8), digest_size=8).hexdigest()

        if world > 1:
            host_digests = [None] * world
            dist.all_gather_object(host_digests, digest(host_final))
        else:
            host_digests = [digest(host_final[f:f + c]) for f, c in (shard_range(n, args.hash_shards, i)
                                                                      for i in range(args.hash_shards))]

    extra = None
    if rank == 0 and world == 1 and want_final:
        try:
            extra = hbm_bound_kernels(eng, S, final, mode)
        except Exception as ex:  # never lose the headline line to an optional measurement
            extra = {"error": str(ex)}

    stats = None
    if want_stats and last is not None:
        if use_dist:
            stats = stats_from_bytes(merge_stats_bytes(records_from_tensor(records)))
        else:
            stats = eng.read_stats(last.stats_raw)
        assert stats.count == n_all, (stats.count, n_all)

    if rank == 0:
        value = n_all * args.steps / dt
        k_s = kernel_ms / 1e3  # summed kernel time of this rank's launches in the timed region
        k_avg_s = k_s / max(launches, 1)
        paths_per_launch = n * args.steps / max(launches, 1)  # host pipeline: one launch per 2^22-path chunk
        writes_final = want_final or to_host
        bytes_per_launch = 4.0 * paths_per_launch if writes_final else 0.0
        achieved = bytes_per_launch / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        traffic, traffic_src = pmc_traffic(mode_name if args.stream != "ref" else "ref", n, periods, outputs)
        insts = VALU_INSTS_PER_STEP[mode_name]
        kind = eng.divide_kind(sim)
        kernel_name = "paths_kernel"
        if args.stream == "ref":
            windowed = periods <= REF_TREE_MAX
            kernel_name = ("ref_windowed_kernel" if periods <= REF_WINDOW[2] else
                           "ref_tree_kernel" if windowed else "ref_generic_kernel")
            # the generic kernel (longer paths still) keeps its generator states in memory and is not priced here
            insts = None
            if windowed and periods and kind in (0, 2):  # fast / checked; the IEEE divide is not counted
                insts = ref_valu_per_path(periods) / periods + (REF_CHECK_VALU / 8 if kind == 2 else 0.0)
        elif args.stream == "2":
            insts = None  # round 1's stream: counts in DESIGN.md section 5, not tracked by the ISA test
        elif kind == 2:  # checked
            insts = insts + VALU_CHECK_PER_STEP[mode_name]
        elif kind == 1:  # IEEE divide: not counted
            insts = None
        valu_ach = paths_per_launch * periods * insts / k_avg_s if k_avg_s > 0 and insts else 0.0
        # the same loop priced with this round's measured issue cost of every opcode (half-rate multiplies,
        # conversions, SDWA ...: tools/valu_model.py over profiles/r04/ubench_ops.jsonl), against the kernel's
        # time at the clock the chip held while it ran -- sampled live by the launches' own workgroups
        model, model_src = (pmc_traffic(mode_name, n, periods, outputs, want="valu") if args.stream == "3" and kind == 0
                            else (None, None))
        weighted_frac = weighted_frac_measured = None
        if model and k_avg_s > 0 and held_clock_ghz > 0:
            simds = eng.geometry()[2] * 4
            blocks_per_simd = paths_per_launch * periods / model["periods_per_block"] / 64.0 / simds
            weighted_frac = model["class_clk_per_block"] * blocks_per_simd / (k_avg_s * held_clock_ghz * 1e9)
            weighted_frac_measured = model["model_clk_per_block"] * blocks_per_simd / (k_avg_s * held_clock_ghz * 1e9)
        if world == 1:
            par = "single GPU"
        elif to_host:
            par = f"path-range shards x{world}, no collective (each rank streams its shard to its own host buffer)"
        else:
            par = f"path-range shards x{world}, one {'RCCL' if args.backend == 'nccl' else 'gloo'} all_gather of the stats record per step"
        workload = (preset["name"] if is_preset else
                    f"{mode_name} returns, {periods} periods x "
                    + (f"{total_paths:.3g} paths in total" if total_paths is not None else f"{per_gpu:.3g} paths per GPU")
                    + f", outputs={outputs}"
                    + (", the reference CPU engine's own stream (per-path mt19937 + libstdc++ Lemire map)"
                       if args.stream == "ref" else f", counter stream v{args.stream}"))
        out = {
            "metric": "simulated paths/sec at N=360 periods" if periods == 360
                      else f"simulated paths/sec at N={periods} periods",
            "value": value, "unit": "paths/s", "n_gpus": world, "ranks": dist.get_world_size() if use_dist else 1,
            "backend": ("rccl (torch nccl)" if args.backend == "nccl" else "gloo") if use_dist else "none",
            "devices": devices, "launcher": "self" if os.environ.get("SMMC_BENCH_CHILD") else
                                            ("external" if have_ranks else "none"),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if total_paths is not None else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "build_digest": _build_digest(),
            "config": {"workload": workload, "baseline_config": args.config if is_preset else None,
                       "paths_rank0": n, "paths_all_ranks": n_all, "n_periods": periods, "mode": mode_name,
                       "outputs": outputs, "seed": hex(SEED), "stream": args.stream,
                       "divide": ("fast", "exact", "checked")[kind],
                       "parallelism": par},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel_name, "kernel_ms": k_avg_s * 1e3, "launches": launches,
                         "bytes_per_launch": bytes_per_launch,
                         "note": "VALU-bound kernel: 4 B of HBM traffic per path by construction; see valu"},
            "valu": {"bound": "valu-issue", "achieved": valu_ach, "peak": VALU_PEAK_LANEOPS, "unit": "lane-ops/s",
                     "frac": valu_ach / VALU_PEAK_LANEOPS, "insts_per_path_period": insts,
                     # the roofline's x axis: VALU lane-operations per algorithmic HBM byte of this launch
                     # (4 bytes written per path), against the ridge point peak ops / peak bytes
                     "arithmetic_intensity": (periods * insts / 4.0) if (insts and writes_final) else None,
                     "ridge_point": VALU_PEAK_LANEOPS / (HBM_PEAK_GBS * 1e9),
                     "weighted_frac": weighted_frac, "weighted_frac_measured_costs": weighted_frac_measured,
                     "weighted_model": ({"class_clk_per_block": model["class_clk_per_block"],
                                         "half_rate_insts_per_block": model["half_rate_insts_per_block"],
                                         "measured_cost_clk_per_block": model["model_clk_per_block"],
                                         "periods_per_block": model["periods_per_block"],
                                         "valu_insts_per_block": model["valu_insts_per_block"], "source": model_src}
                                        if model else None),
                     "held_clock_ghz": held_clock_ghz or None,
                     "held_clock_source": ("live: s_memtime / s_memrealtime of every workgroup of the timed launches "
                                           "(smmc_engine_kernel_clock)") if held_clock_ghz else None,
                     "note": "frac is unweighted: every VALU instruction counted as one 2-clock issue slot at the 2.4 GHz peak; "
                             "weighted_frac prices every opcode at its class cost (2 clocks full rate, 4 half rate: the "
                             "classes measured by tools/ubench_ops.hip) and uses the clock the chip held; "
                             "weighted_frac_measured_costs uses the probes' measured costs, which carry their own overhead; arithmetic_intensity in lane-ops per HBM byte"},
        }
        if to_host:
            out["host_pipeline"] = {"bytes_to_host_per_step": 4.0 * n, "GBps_rank0": 4.0 * n * args.steps / dt / 1e9,
                                    "kernel_busy_frac": k_s / dt if dt > 0 else None, "pinned": True}
        if host_digests is not None:
            out["host_digests"] = host_digests
        if stats is not None:
            out["result"] = {"mean": stats.mean, "std": stats.std, "below_initial": stats.below,
                             "hist_total": int(stats.hist.sum()) + stats.underflow + stats.overflow}
        if extra is not None:
            out["hbm_bound_kernels"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(table)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world > 1 and not os.environ.get("SMMC_BENCH_CHILD") and not os.environ.get("SMMC_BENCH_NO_GROUP"):
            # ranks made by an external launcher (torch.distributed.run): there is no GPU-less parent of ours, so
            # rank 0 -- its timed region over, its collectives done, its engine closed -- starts the fresh child
            # that runs the workload through the one-process C-ABI launcher (self-launched runs: the parent does)
            eng.close()
            del final
            torch.cuda.empty_cache()
            out["group_single_process"] = run_group_child(world, sys.argv[1:])
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
