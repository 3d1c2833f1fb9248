"""Host-side mirror of the reference's Monte-Carlo API on top of the C ABI.

Function names, argument meaning and error behaviour follow
include/stock_market_monte_carlo/simulations.h of the reference (cited per function);
the work happens in libsmmc_hip.so on the MI355X.  PyTorch only provides device
memory and the stream.
"""
import ctypes as C
import dataclasses
import os

import numpy as np

from . import _lib
from ._lib import MODE_GAUSSIAN, MODE_TABLE, SmmcError

DEFAULT_TABLE_CSV = "data/SP500_monthly_returns.csv"  # examples/benchmark_mc_cpu_v2.cpp:25


@dataclasses.dataclass
class Stats:
    """Host copy of a packed statistics record (smmc_stats + bucket counts)."""
    count: int
    below: int
    underflow: int
    overflow: int
    sum: float
    sumsq: float
    min: float
    max: float
    hist: np.ndarray

    @property
    def mean(self):
        return self.sum / self.count if self.count else float("nan")

    @property
    def std(self):
        """Population standard deviation (examples/benchmark_mc_gpu.cpp:19-27)."""
        if not self.count:
            return float("nan")
        m = self.mean
        return max(self.sumsq / self.count - m * m, 0.0) ** 0.5


def stats_from_bytes(raw):
    """raw: bytes/uint8 array holding one packed record."""
    buf = np.frombuffer(bytes(raw), dtype=np.uint8)
    hdr = _lib.Stats.from_buffer_copy(buf[: C.sizeof(_lib.Stats)].tobytes())
    hist = np.frombuffer(buf[C.sizeof(_lib.Stats):].tobytes(), dtype=np.uint64)[: hdr.n_bins].copy()
    return Stats(hdr.count, hdr.below, hdr.underflow, hdr.overflow, hdr.sum, hdr.sumsq, hdr.min, hdr.max, hist)


def merge_stats_bytes(records):
    """Merges packed records (same n_bins) in the given order; returns bytes."""
    L = _lib.lib()
    acc = bytearray(records[0])
    dst = (C.c_char * len(acc)).from_buffer(acc)
    for r in records[1:]:
        src = (C.c_char * len(r)).from_buffer_copy(bytes(r))
        _lib.check(L.smmc_stats_merge(dst, src))
    return bytes(acc)


@dataclasses.dataclass
class SimResult:
    final: object = None        # torch.float32 [n_paths] on the engine's device, or None
    chunk_mean: object = None   # torch.float32 [ceil(n/256)] or None
    chunk_var: object = None
    stats_raw: object = None    # torch.uint8 [smmc_stats_bytes(n_bins)] on device, or None


class Engine:
    """One engine per (process, device): table, workspace and stream stay resident."""

    def __init__(self, device=0, stream="torch"):
        import torch
        self._torch = torch
        self._L = _lib.lib()
        if not torch.cuda.is_available():
            raise SmmcError("no MI355X visible to this process; the engine has no CPU fallback")
        self.device = int(device)
        self.tdevice = torch.device("cuda", self.device)
        # "torch": launch on torch's current stream of that device (handle 0 = the default
        # stream), so tensors produced here are ordered with the caller's torch work;
        # "new": an engine-owned non-blocking stream; or a raw hipStream_t handle.
        if stream == "torch":
            sp = int(torch.cuda.current_stream(self.tdevice).cuda_stream)
        elif stream == "new":
            sp = -1  # SMMC_STREAM_NEW
        else:
            sp = int(stream)
        self.own_stream = sp == -1
        self.follow_torch = stream == "torch"
        h = C.c_void_p()
        _lib.check(self._L.smmc_engine_create(self.device, C.c_void_p(sp), C.byref(h)))
        self._h = h
        self.table_len = 0

    # -- stream discipline ---------------------------------------------------
    def _enter(self):
        """Before enqueuing: launches go to the caller's CURRENT torch stream (a "torch" engine
        re-binds on every call: the caller may be inside `with torch.cuda.stream(s)` now), or the
        engine's own stream first waits for it (its outputs were just allocated there)."""
        cur = int(self._torch.cuda.current_stream(self.tdevice).cuda_stream)
        if self.follow_torch:
            _lib.check(self._L.smmc_engine_set_stream(self._h, C.c_void_p(cur)))
        elif self.own_stream:
            _lib.check(self._L.smmc_engine_wait_stream(self._h, C.c_void_p(cur)))
        return cur

    def _leave(self, cur, *tensors):
        """After enqueuing on an engine-owned stream: torch's current stream waits for the engine's
        work, so whatever torch does next with the outputs there -- read, free, reuse the block --
        is ordered after the kernels that wrote them (no record_stream: the caching allocator would
        poll events on a stream the engine may already have destroyed)."""
        if self.own_stream:
            _lib.check(self._L.smmc_engine_release_to_stream(self._h, C.c_void_p(cur)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.smmc_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration -------------------------------------------------------
    def set_table(self, returns_percent):
        t = np.ascontiguousarray(returns_percent, dtype=np.float32)
        self._enter()  # never act on a stream bound by an earlier call: the caller may have destroyed it since
        _lib.check(self._L.smmc_engine_set_table(self._h, t.ctypes.data_as(C.c_void_p), t.size))
        self.table_len = int(t.size)

    def geometry(self):
        g, b, cu = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _lib.check(self._L.smmc_engine_geometry(self._h, C.byref(g), C.byref(b), C.byref(cu)))
        return g.value, b.value, cu.value

    def timing(self, enable=True):
        _lib.check(self._L.smmc_engine_timing(self._h, 1 if enable else 0))

    def kernel_ms(self):
        """(summed main-kernel milliseconds, launches) since the last call; synchronises."""
        ms, n = C.c_double(), C.c_uint32()
        self._enter()
        _lib.check(self._L.smmc_engine_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_clock_ghz(self):
        """The shader clock the chip held, averaged over the workgroups of the path-kernel launches timed since the
        last call (0.0 if none was sampled); synchronises."""
        ghz = C.c_double()
        self._enter()
        _lib.check(self._L.smmc_engine_kernel_clock(self._h, C.byref(ghz)))
        return ghz.value

    def selftest(self, bits_lo, bits_hi):
        """Mismatches of the device's divide-by-100 shortcut vs the IEEE divide on [lo, hi)."""
        a = C.c_uint64()
        _lib.check(self._L.smmc_engine_selftest(self._h, bits_lo, bits_hi, C.byref(a)))
        return a.value

    def sync(self):
        """Waits for everything the engine has enqueued.  A "torch" engine first re-binds to the caller's
        current stream (work enqueued on the stream of an earlier call stays ordered before it; a stream
        handed to the engine must outlive its pending work, not the engine)."""
        self._enter()
        _lib.check(self._L.smmc_engine_sync(self._h))

    # -- simulation ----------------------------------------------------------
    @staticmethod
    def make_sim(n_paths, n_periods, mode, seed, first_path=0, initial_capital=1000.0, gauss_mean=0.5,
                 gauss_std=0.83333, n_bins=0, hist_lo=0.0, hist_hi=1.0, below_threshold=None,
                 exact_div=False, stream=3):
        s = _lib.Sim()
        s.struct_size = C.sizeof(_lib.Sim)
        s.mode = mode
        s.seed = seed & 0xFFFFFFFFFFFFFFFF
        s.first_path = first_path
        s.n_paths = n_paths
        s.n_periods = n_periods
        s.initial_capital = initial_capital
        s.gauss_mean = gauss_mean
        s.gauss_std = gauss_std
        s.n_bins = n_bins
        s.hist_lo = hist_lo
        s.hist_hi = hist_hi
        s.below_threshold = initial_capital if below_threshold is None else below_threshold
        if stream not in (2, 3, "ref"):
            raise ValueError("stream must be 3 or 2 (the counter stream: Philox counter layout in both modes and "
                             "the Gaussian draw) or 'ref' (the reference CPU engine's per-path mt19937 stream)")
        s.flags = ((_lib.FLAG_EXACT_DIV if exact_div else 0) | (_lib.FLAG_STREAM_V2 if stream == 2 else 0)
                   | (_lib.FLAG_STREAM_REF if stream == "ref" else 0))
        return s

    def simulate(self, sim, want_final=True, want_chunk_stats=False, want_stats=False, out=None):
        """Enqueues one simulation on the engine stream; returns device tensors."""
        torch = self._torch
        n = int(sim.n_paths)
        res = SimResult()
        if want_final:
            res.final = out if out is not None else torch.empty(n, dtype=torch.float32, device=self.tdevice)
            assert res.final.numel() >= n and res.final.dtype == torch.float32 and res.final.is_contiguous()
        if want_chunk_stats:
            nc = (n + _lib.CHUNK - 1) // _lib.CHUNK
            res.chunk_mean = torch.empty(nc, dtype=torch.float32, device=self.tdevice)
            res.chunk_var = torch.empty(nc, dtype=torch.float32, device=self.tdevice)
        if want_stats:
            res.stats_raw = torch.empty(int(self._L.smmc_stats_bytes(sim.n_bins)), dtype=torch.uint8,
                                        device=self.tdevice)
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None  # noqa: E731
        cur = self._enter()
        _lib.check(self._L.smmc_engine_simulate(self._h, C.byref(sim), ptr(res.final), ptr(res.chunk_mean),
                                                ptr(res.chunk_var), ptr(res.stats_raw)))
        self._leave(cur, res.final, res.chunk_mean, res.chunk_var, res.stats_raw)
        return res

    def read_stats(self, stats_raw):
        """Copies a device record to the host after the engine stream has drained."""
        self.sync()
        return stats_from_bytes(stats_raw.cpu().numpy().tobytes())

    def stream_handle(self):
        """The hipStream_t (as an int) launches currently go to."""
        raw = C.c_void_p()
        _lib.check(self._L.smmc_engine_get_stream(self._h, C.byref(raw)))
        return int(raw.value or 0)

    # -- statistics of values already in HBM ---------------------
    def _check_values(self, values):
        torch = self._torch
        if not (values.is_cuda and values.dtype == torch.float32 and values.is_contiguous() and values.dim() == 1):
            raise ValueError("values must be a contiguous 1-D float32 tensor on the engine's device")

    def values_stats(self, values, below_threshold=1000.0, n_bins=0, hist_lo=0.0, hist_hi=1.0):
        """One HBM pass over a device tensor -> packed statistics record (device uint8 tensor)."""
        self._check_values(values)
        rec = self._torch.empty(int(self._L.smmc_stats_bytes(n_bins)), dtype=self._torch.uint8, device=self.tdevice)
        cur = self._enter()
        _lib.check(self._L.smmc_engine_values_stats(self._h, C.c_void_p(values.data_ptr()), values.numel(),
                                                    below_threshold, n_bins, hist_lo, hist_hi,
                                                    C.c_void_p(rec.data_ptr())))
        self._leave(cur, rec, values)
        return rec

    def order_statistics(self, values, ranks):
        """Exact k-th smallest values (0-based ranks) of a device tensor, unsorted input."""
        self._check_values(values)
        r = np.ascontiguousarray(ranks, dtype=np.uint64)
        out = np.empty(r.size, dtype=np.float32)
        self._leave(self._enter(), values)  # synchronous call: only the input needs ordering
        _lib.check(self._L.smmc_engine_order_statistics(self._h, C.c_void_p(values.data_ptr()), values.numel(),
                                                        r.ctypes.data_as(C.c_void_p), r.size,
                                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def quartiles(self, values):
        """{min, Q1, Q2, Q3, max}: update_quartiles, examples/visualize_returns_cpu_v2.cpp:83-111."""
        self._check_values(values)
        out = np.empty(5, dtype=np.float32)
        self._leave(self._enter(), values)
        _lib.check(self._L.smmc_engine_quartiles(self._h, C.c_void_p(values.data_ptr()), values.numel(),
                                                 out.ctypes.data_as(C.c_void_p)))
        return out

    def reduce_mean_host(self, host_values):
        v = np.ascontiguousarray(host_values, dtype=np.float32)
        mean, total = C.c_float(), C.c_double()
        _lib.check(self._L.smmc_engine_reduce_mean_host(self._h, v.ctypes.data_as(C.c_void_p), v.size,
                                                        C.byref(mean), C.byref(total)))
        return mean.value, total.value

    def divide_kind(self, sim, keepdata=False):
        """DIV_FAST / DIV_EXACT / DIV_CHECKED: which divide-by-100 a launch of `sim` uses (results
        never depend on it)."""
        rc = self._L.smmc_engine_divide_kind(self._h, C.byref(sim), 1 if keepdata else 0)
        if rc < 0:
            _lib.check(rc)
        return rc

    def simulate_keepdata(self, sim, want_final=True):
        torch = self._torch
        n, p = int(sim.n_paths), int(sim.n_periods)
        traj = torch.empty((n, p + 1), dtype=torch.float32, device=self.tdevice)
        final = torch.empty(n, dtype=torch.float32, device=self.tdevice) if want_final else None
        if n:
            cur = self._enter()
            _lib.check(self._L.smmc_engine_simulate_keepdata(
                self._h, C.byref(sim), C.c_void_p(traj.data_ptr()),
                C.c_void_p(final.data_ptr()) if final is not None else None))
            self._leave(cur, traj, final)
        return traj, final

    def simulate_to_host(self, sim, out=None, want_stats=False, want_chunk_stats=False, progress=None):
        """Final values (and optionally per-256-path means/variances) straight into host
        memory: chunked, D2H overlapped with compute.  Returns (final, stats, (means, vars))."""
        n = int(sim.n_paths)
        host = out if out is not None else np.empty(n, dtype=np.float32)
        assert host.dtype == np.float32 and host.size >= n and host.flags.c_contiguous
        st = _lib.Stats()
        hist = np.zeros(max(int(sim.n_bins), 1), dtype=np.uint64)
        nc = (n + _lib.CHUNK - 1) // _lib.CHUNK
        cm = np.empty(nc, dtype=np.float32) if want_chunk_stats else None
        cv = np.empty(nc, dtype=np.float32) if want_chunk_stats else None
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
        self._enter()  # synchronous: returns with both of its streams drained
        _lib.check(self._L.smmc_engine_simulate_to_host(
            self._h, C.byref(sim), vp(host), vp(cm), vp(cv), C.byref(progress) if progress is not None else None,
            C.byref(st) if want_stats else None, vp(hist) if want_stats else None))
        stats = None
        if want_stats:
            stats = Stats(st.count, st.below, st.underflow, st.overflow, st.sum, st.sumsq, st.min, st.max,
                          hist[: int(sim.n_bins)])
        return host, stats, (cm, cv)

    def simulate_keepdata_to_host(self, sim):
        n, p = int(sim.n_paths), int(sim.n_periods)
        traj = np.empty((n, p + 1), dtype=np.float32)
        final = np.empty(n, dtype=np.float32)
        self._enter()
        _lib.check(self._L.smmc_engine_simulate_keepdata_to_host(
            self._h, C.byref(sim), traj.ctypes.data_as(C.c_void_p), final.ctypes.data_as(C.c_void_p)))
        return traj, final


class Group:
    """Several devices of this process behind one call (smmc_group_*, include/smmc.h): the request is
    sharded by contiguous global path ids, every device streams its share to its place in the host
    arrays, and ONE merged statistics record comes back -- merged on the host in device order
    (merge="host") or by one RCCL all-reduce of the integer fields (merge="rccl", distinct devices).
    Reference: mc_simulations_multi_gpu_launcher_async, src/simulations.cu:576-655."""

    def __init__(self, devices, merge="host"):
        self._L = _lib.lib()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        kind = {"host": _lib.MERGE_HOST, "rccl": _lib.MERGE_RCCL}[merge]
        _lib.check(self._L.smmc_group_create(devs, len(devices), kind, C.byref(h)))
        self._h = h
        self.merge = merge

    def close(self):
        if getattr(self, "_h", None):
            self._L.smmc_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._L.smmc_group_size(self._h))

    def set_table(self, returns_percent):
        t = np.ascontiguousarray(returns_percent, dtype=np.float32)
        _lib.check(self._L.smmc_group_set_table(self._h, t.ctypes.data_as(C.c_void_p), t.size))

    def shard(self, n_paths, index):
        first, count = C.c_uint64(), C.c_uint64()
        _lib.check(self._L.smmc_group_shard(self._h, n_paths, index, C.byref(first), C.byref(count)))
        return first.value, count.value

    def simulate(self, sim, out=None, want_final=True, want_stats=False, want_chunk_stats=False, progress=None):
        """Returns (final or None, Stats or None, (chunk means, chunk variances))."""
        n = int(sim.n_paths)
        host = None
        if want_final:
            host = out if out is not None else np.empty(n, dtype=np.float32)
            assert host.dtype == np.float32 and host.size >= n and host.flags.c_contiguous
        st = _lib.Stats()
        hist = np.zeros(max(int(sim.n_bins), 1), dtype=np.uint64)
        nc = (n + _lib.CHUNK - 1) // _lib.CHUNK
        cm = np.empty(nc, dtype=np.float32) if want_chunk_stats else None
        cv = np.empty(nc, dtype=np.float32) if want_chunk_stats else None
        vp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731
        _lib.check(self._L.smmc_group_simulate(
            self._h, C.byref(sim), vp(host), vp(cm), vp(cv), C.byref(progress) if progress is not None else None,
            C.byref(st) if want_stats else None, vp(hist) if want_stats else None))
        stats = None
        if want_stats:
            stats = Stats(st.count, st.below, st.underflow, st.overflow, st.sum, st.sumsq, st.min, st.max,
                          hist[: int(sim.n_bins)])
        return host, stats, (cm, cv)

    def device_record(self, index):
        """merge="rccl": device pointer (int) of device `index`'s copy of the merged packed record."""
        p = C.c_void_p()
        _lib.check(self._L.smmc_group_device_record(self._h, index, C.byref(p)))
        return int(p.value or 0)

    def timings(self):
        """(engines up, communicator init, merge step of the last simulate) in milliseconds."""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        _lib.check(self._L.smmc_group_timings(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value


# ---------------------------------------------------------------------------
# Reference-named functions (include/stock_market_monte_carlo/simulations.h)
# ---------------------------------------------------------------------------

_engines = {}


def _engine(device=0, lane=0):
    """One engine per (device, lane); lane > 0 only when SMMC_DEVICE_MAP puts several shards of one
    call on the same device."""
    e = _engines.get((device, lane))
    if e is None:
        e = _engines[(device, lane)] = Engine(device)
    return e


def _device_map(n_gpus):
    """Shard g of an n_gpus-way call runs on device map[g]: g itself, or SMMC_DEVICE_MAP="0,0,1"
    (the same hook as the C++ layer's, csrc/smmc_dropin.cpp)."""
    import torch
    env = os.environ.get("SMMC_DEVICE_MAP")
    devs = [int(x) for x in env.split(",")][:n_gpus] if env else list(range(n_gpus))
    if len(devs) < n_gpus:
        raise ValueError("SMMC_DEVICE_MAP names fewer devices than n_gpus")
    have = torch.cuda.device_count()
    if n_gpus < 1 or any(d < 0 or d >= have for d in devs):
        raise ValueError(f"n_gpus={n_gpus} but {have} device(s) visible")
    return devs


def _seed(seed):
    # the reference seeds every path from std::random_device (src/simulations.cpp:245-246)
    return int.from_bytes(os.urandom(8), "little") if seed is None else int(seed)


def update_fund(fund_value, period_return):
    """simulations.h:9, src/simulations.cpp:14-16."""
    return float(_lib.lib().smmc_update_fund(float(fund_value), float(period_return)))


def many_updates(fund_value, returns, n_periods):
    """simulations.h:11-13, src/simulations.cpp:24-39: n_periods + 1 values, [0] = fund_value."""
    r = np.ascontiguousarray(returns, dtype=np.float32)
    if r.size < n_periods:
        raise ValueError("returns holds fewer than n_periods entries")
    out = np.empty(n_periods + 1, dtype=np.float32)
    out[0] = np.float32(fund_value)
    _lib.lib().smmc_many_updates(r.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), n_periods)
    return out


def read_historical_returns(csv_fpath):
    """simulations.h:31, src/simulations.cpp:83-93: the `returns` column of a CSV, percent units.
    Rows whose cell is empty (first month of python/get_data.py:59 output) are skipped."""
    vals = []
    with open(csv_fpath) as f:
        header = [h.strip() for h in f.readline().rstrip("\r\n").split(",")]
        if "returns" not in header:
            raise ValueError(f"{csv_fpath}: no column named 'returns'")
        col = header.index("returns")
        for line in f:
            cells = line.rstrip("\r\n").split(",")
            if col < len(cells) and cells[col].strip() not in ("", "nan", "NaN"):
                vals.append(np.float32(cells[col]))
    return np.array(vals, dtype=np.float32)


_groups = {}


def _group(devices):
    """One cached Group (smmc_group_*, host merge) per device list, like the C++ layer's."""
    key = tuple(devices)
    g = _groups.get(key)
    if g is None:
        g = _groups[key] = Group(list(devices))
    return g


def mc_simulations_gpu(max_n_simulations, n_periods, initial_capital, returns, n_gpus=1, seed=None, stream=3):
    """simulations.h:73-79, src/simulations.cu:661-680: final value of every path (host array).
    stream: 3 (default) / 2 the build's counter streams; "ref" the reference CPU engine's own stream --
    path id draws from mt19937(seed + id) through libstdc++'s uniform_int_distribution.
    Paths shard over n_gpus devices of this process by contiguous global id ranges through the C entry
    for several devices (smmc_group_simulate: one host thread per device, so that all devices compute and
    copy at once -- the reference's async launcher, src/simulations.cu:599-626; the N mod G remainder is
    kept).  SMMC_DEVICE_MAP="0,0,1" places shard g on device map[g]."""
    devs = _device_map(n_gpus)
    n = int(max_n_simulations)
    g = _group(devs)
    g.set_table(returns)
    sim = Engine.make_sim(n, n_periods, MODE_TABLE, _seed(seed), initial_capital=initial_capital, stream=stream)
    out, _, _ = g.simulate(sim)
    return out


def mc_simulations(max_n_simulations, n_periods, initial_capital, historical_returns, final_values=None, seed=None,
                   stream=3):
    """simulations.h:49-54, src/simulations.cpp:204-266 (the CPU v2 engine), run on the GPU.
    With stream="ref" and seed=s the result is, bit for bit, what that engine computes when the
    std::random_device of path id returns s + id (src/simulations.cpp:245-246).
    final_values, if given, must be pre-sized like the reference's caller does
    (examples/benchmark_mc_cpu_v2.cpp:26) and is filled in place."""
    n = int(max_n_simulations)
    if final_values is not None and (final_values.size < n or final_values.dtype != np.float32):
        raise ValueError("final_values must be a float32 array of at least max_n_simulations entries")
    e = _engine(0)
    e.set_table(historical_returns)
    sim = Engine.make_sim(n, n_periods, MODE_TABLE, _seed(seed), initial_capital=initial_capital, stream=stream)
    out, _, _ = e.simulate_to_host(sim, out=final_values)
    return out[:n]


def mc_simulations_gpu_reduceBlock(max_n_simulations, n_periods, initial_capital, returns, n_gpus=1, seed=None):
    """simulations.h:81-88, src/simulations.cu:682-697: (means, variances), one pair per 256 paths."""
    if n_gpus != 1:
        # src/simulations.cu:693 throws std::invalid_argument
        raise ValueError("mc_simulations_gpu_reduceBlock supports n_gpus == 1 only")
    e = _engine(0)
    e.set_table(returns)
    sim = Engine.make_sim(int(max_n_simulations), n_periods, MODE_TABLE, _seed(seed), initial_capital=initial_capital)
    r = e.simulate(sim, want_final=False, want_chunk_stats=True)
    e.sync()
    return r.chunk_mean.cpu().numpy(), r.chunk_var.cpu().numpy()


def mc_simulations_keepdata(max_n_simulations, n_periods, initial_capital, historical_returns, seed=None):
    """simulations.h:57-63, src/simulations.cpp:139-202: (mc_data [N, P+1], final_values)."""
    e = _engine(0)
    e.set_table(historical_returns)
    sim = Engine.make_sim(int(max_n_simulations), n_periods, MODE_TABLE, _seed(seed), initial_capital=initial_capital)
    return e.simulate_keepdata_to_host(sim)


def reduce_mean_gpu(vec, n=None):
    """simulations.h:71, src/simulations.cu:269-341: mean of the first n entries of a host array."""
    v = np.ascontiguousarray(vec, dtype=np.float32)
    n = v.size if n is None else int(n)
    return _engine(0).reduce_mean_host(v[:n])[0]


def vector_add_gpu(a, b):
    """out = a + b for two host float arrays on the MI355X (src/gpu.cu:17-47, include/.../gpu.h:2; the
    demo north_star names beside the engine).  Returns (out, seconds of the launch alone)."""
    x = np.ascontiguousarray(a, dtype=np.float32)
    y = np.ascontiguousarray(b, dtype=np.float32)
    if x.shape != y.shape or x.ndim != 1:
        raise ValueError("vector_add_gpu takes two 1-D arrays of one length")
    out = np.empty_like(x)
    sec = C.c_double()
    _lib.check(_lib.lib().smmc_vector_add(out.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                                          y.ctypes.data_as(C.c_void_p), x.size, C.byref(sec)))
    return out, sec.value


def _to_device(values, n_el):
    import torch
    v = np.ascontiguousarray(values, dtype=np.float32)[: int(n_el)]
    return torch.from_numpy(v).to(_engine(0).tdevice)


def update_quartiles(vec, n_el):
    """examples/visualize_returns_cpu_v2.cpp:83-111: [min, Q1, Q2, Q3, max] of vec[:n_el]."""
    return _engine(0).quartiles(_to_device(vec, n_el))


def update_mean_std(v, n_el):
    """examples/visualize_returns_cpu_v2.cpp:113-123: (mean, population std) as floats."""
    e = _engine(0)
    st = e.read_stats(e.values_stats(_to_device(v, n_el)))
    # variance from the DOUBLE mean, clamped, then rounded: with the float-rounded mean the error
    # 2 * mean * ulp(mean) swamps a small variance (the reference sums (v - mean)^2 instead)
    m = st.sum / n_el
    return float(np.float32(m)), float(np.float32(np.sqrt(max(st.sumsq / n_el - m * m, 0.0))))


def update_count_below_min(min_final_amount, final_values, n_simulations):
    """examples/visualize_returns_cpu_v2.cpp:125-138: count of final_values[:n] < min_final_amount."""
    e = _engine(0)
    return e.read_stats(e.values_stats(_to_device(final_values, n_simulations), below_threshold=min_final_amount)).below
