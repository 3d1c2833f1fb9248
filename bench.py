#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Monte-Carlo returns engine.

Metric (BASELINE.json): simulated paths/sec at 360 periods.  One "step" = one pass
of the hot path over one batch of synthetic input: PATHS_PER_GPU paths x 360 periods
per GPU, Gaussian returns (BASELINE configs[1]: "Gaussian returns, 360 periods x 1e8
paths, 1 x MI355X, final-value only + block-reduce mean"), producing the final value
of every path in HBM, the per-256-path block means/variances and the fused
statistics record (sum, sum of squares, below-count, 100-bucket histogram).

Multi-GPU (`torchrun`-style launch, one rank per GPU): paths shard by contiguous
global id ranges (weak scaling: PATHS_PER_GPU each); the only exchange is one RCCL
all_gather of the ~900-byte statistics record per step, merged in rank order.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline      HBM roofline of the dominant kernel (paths_kernel): algorithmic bytes
                (4 B per path, the coalesced final-value store) / its HIP-event time.
                The kernel is VALU-bound by design, so this fraction is tiny; the
                `valu` object carries the roof that actually binds.
  cpu_baseline  the reference's CPU algorithm (oracle engine R: mt19937 + Lemire +
                update_fund, OpenMP hw-1 threads) timed on this host on a bounded sample.
"""
import argparse
import json
import os
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
# VALU work per path-period of paths_kernel's inner loop, from the gfx950 ISA
# (tools/isa_loop_count.py): instructions, and issue units where a
# plain VALU op = 1 and multi-cycle ones carry their measured cost (mad_u64 2.29, ...)
# HBM bytes per launch of paths_kernel from the PMC passes committed in
# profiles/r01/pmc_summary.txt (WRITE_SIZE KiB + 2 x FETCH_SIZE KiB: the gfx950 read
# counter tallies 128-byte requests at 64 bytes), measured on the default workload only
# (1e8 paths, outputs=all): bench.py cannot run rocprofv3 on itself.
PMC_TRAFFIC_BYTES = {"gaussian": (421446 + 2 * 134) * 1024, "table": (429119 + 2 * 97) * 1024}
VALU_INSTS_PER_STEP = {"gaussian": 122 / 4, "table": 96 / 8}
VALU_UNITS_PER_STEP = {"gaussian": 37.8, "table": 17.7}
VALU_CHECK_PER_STEP = {"gaussian": 2 / 4, "table": 2 / 8}  # the range-checked divide: two compares per Philox block


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


def cpu_baseline(table, budget_s=15.0):
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
        _, used = O.ref_mc_simulations(n, N_PERIODS, 1000.0, table, 12345, n_threads=threads)
        dt = time.perf_counter() - t0
        if dt >= budget_s * 2 / 3 or n >= 200_000_000:
            break
        n = int(n * min(max(budget_s / max(dt, 1e-3), 1.5), 20.0))
        n -= n % 1000
    # the same engine on ONE thread (SURVEY 8d asks for both figures): ~3 s
    n1 = max(1000, int(n / dt * 3.0 / max(used, 1)))
    n1 -= n1 % 1000
    t0 = time.perf_counter()
    O.ref_mc_simulations(n1, N_PERIODS, 1000.0, table, 12345, n_threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": n / dt, "unit": "paths/s", "cores": used, "kind": "port",
            "single_thread_value": n1 / dt1,
            "sample": f"{n} paths x {N_PERIODS} periods, table mode (T={table.size}), oracle engine R "
                      f"(per-path mt19937 + Lemire + update_fund, OpenMP dynamic blocks of 1000, deterministic "
                      f"seeds: no per-path std::random_device), {dt:.1f} s; single thread: {n1} paths, {dt1:.1f} s"}


def hbm_bound_kernels(eng, S, final, mode):
    """The path's HBM-bound neighbours, timed live with HIP events on the
    engine's stream AFTER the timed region: statistics and exact quartiles of the resident final
    values (4 B read per value per pass), and keepdata (4 (P+1) B written per path)."""
    import ctypes as C
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
    return out


def _lib_keepdata(eng, sim, traj):
    import ctypes as C
    from stock_market_monte_carlo_amd import _lib
    _lib.check(eng._L.smmc_engine_simulate_keepdata(eng._h, C.byref(sim), C.c_void_p(traj.data_ptr()), None))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["gaussian", "table"], default="gaussian")
    ap.add_argument("--paths-per-gpu", type=int, default=PATHS_PER_GPU)
    ap.add_argument("--periods", type=int, default=N_PERIODS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL (the real multi-GPU run); gloo lets several ranks rehearse on ONE GPU")
    ap.add_argument("--outputs", choices=["all", "final", "stats"], default="all",
                    help="all = final values + block means + statistics (configs[1]); final = final values only; "
                         "stats = statistics only (no per-path HBM write)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import stock_market_monte_carlo_amd as S
    from stock_market_monte_carlo_amd.dist import gather_stats_records
    from stock_market_monte_carlo_amd.engine import merge_stats_bytes, stats_from_bytes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = local_rank % max(torch.cuda.device_count(), 1) if args.backend == "gloo" else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    table = load_table()
    eng = S.Engine(device)
    eng.set_table(table)
    mode = S.MODE_GAUSSIAN if args.mode == "gaussian" else S.MODE_TABLE
    n = args.paths_per_gpu
    sim = S.Engine.make_sim(n, args.periods, mode, SEED, first_path=rank * n, initial_capital=1000.0,
                            gauss_mean=0.5, gauss_std=0.83333, n_bins=100, hist_lo=0.0, hist_hi=20000.0)
    want_final = args.outputs in ("all", "final")
    want_chunks = args.outputs == "all"
    want_stats = args.outputs in ("all", "stats")
    final = torch.empty(n, dtype=torch.float32, device=eng.tdevice) if want_final else None
    records = None

    def step():
        # the whole per-step job: simulate this rank's shard, then (N > 1) the one RCCL
        # all_gather of the statistics records, merged in rank order on the host
        nonlocal records
        r = eng.simulate(sim, want_final=want_final, want_chunk_stats=want_chunks, want_stats=want_stats, out=final)
        if want_stats:
            records = gather_stats_records(r.stats_raw) if world > 1 else None
        return r

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = eng.kernel_ms()
    eng.timing(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=eng.tdevice)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    extra = None
    if rank == 0 and world == 1 and want_final:
        try:
            extra = hbm_bound_kernels(eng, S, final, mode)
        except Exception as ex:  # never lose the headline line to an optional measurement
            extra = {"error": str(ex)}

    stats = None
    if want_stats:
        if world > 1:
            stats = stats_from_bytes(merge_stats_bytes(records))
        else:
            stats = eng.read_stats(last.stats_raw)
        assert stats.count == n * world, (stats.count, n * world)

    if rank == 0:
        total_paths = n * world * args.steps
        value = total_paths / dt
        k_avg_s = kernel_ms / 1e3 / max(launches, 1)
        bytes_per_launch = 4.0 * n if want_final else 0.0
        achieved = bytes_per_launch / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
        default_workload = n == PATHS_PER_GPU and args.periods == N_PERIODS and args.outputs == "all"
        traffic = float(PMC_TRAFFIC_BYTES[args.mode]) if default_workload else None
        insts, units = VALU_INSTS_PER_STEP[args.mode], VALU_UNITS_PER_STEP[args.mode]
        kind = eng.divide_kind(sim)
        if kind == 2:  # checked
            insts, units = insts + VALU_CHECK_PER_STEP[args.mode], units + VALU_CHECK_PER_STEP[args.mode]
        elif kind == 1:  # IEEE divide: not counted (28.3 units per period in table mode)
            insts = units = None
        valu_ach = n * args.periods * insts / k_avg_s if k_avg_s > 0 and insts else 0.0
        valu_w = n * args.periods * units / k_avg_s if k_avg_s > 0 and units else 0.0
        out = {
            "metric": "simulated paths/sec at N=360 periods" if args.periods == 360
                      else f"simulated paths/sec at N={args.periods} periods",
            "value": value, "unit": "paths/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.mode} returns, {args.periods} periods x {n:.3g} paths per GPU, "
                                   f"outputs={args.outputs}"
                                   + (f" (BASELINE configs[{1 if args.mode == 'gaussian' else 2}])" if default_workload else ""),
                       "paths_per_gpu": n, "n_periods": args.periods, "mode": args.mode, "seed": hex(SEED),
                       "divide": ("fast", "exact", "checked")[kind],
                       "parallelism": f"path-range shards x{world}, one RCCL all_gather of the stats record per step"
                                      if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "paths_kernel", "kernel_ms": k_avg_s * 1e3, "bytes_per_launch": bytes_per_launch,
                         "note": "VALU-bound kernel: 4 B of HBM traffic per 360-period path by construction; "
                                 "see valu"},
            "valu": {"bound": "valu-issue", "achieved": valu_ach, "peak": VALU_PEAK_LANEOPS, "unit": "lane-ops/s",
                     "frac": valu_ach / VALU_PEAK_LANEOPS, "insts_per_path_period": insts,
                     "issue_weighted_frac": valu_w / VALU_PEAK_LANEOPS, "issue_units_per_path_period": units},
        }
        if stats is not None:
            out["result"] = {"mean": stats.mean, "std": stats.std, "below_initial": stats.below,
                             "hist_total": int(stats.hist.sum()) + stats.underflow + stats.overflow}
        if extra is not None:
            out["hbm_bound_kernels"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(table)
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
