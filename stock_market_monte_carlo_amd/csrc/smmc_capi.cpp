// smmc_capi.cpp -- the C ABI declared in include/smmc.h: engine lifetime, argument
// checking, kernel-variant selection, streams/events, the host-buffer pipeline.
//
// Host runtime counterpart of the reference launchers (src/simulations.cu:345-697),
// re-designed: a long-lived engine per device instead of cudaMalloc/cudaFree per
// call, every HIP call checked and reported through a return code (the reference
// prints and exit()s, src/simulations.cu:23-30), no device-wide synchronisation.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "smmc_host.h"
#include "smmc_internal.h"

namespace {

#include "smmc_bm_tables.inc"  // smmc_bm_radius[1024][4], smmc_bm_trig[256][2]

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define SMMC_HIP(call)                                                                         \
  do {                                                                                         \
    hipError_t err__ = (call);                                                                 \
    if (err__ != hipSuccess)                                                                   \
      return fail(SMMC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__),     \
                  __FILE__, __LINE__);                                                         \
  } while (0)

// Inside an open timing pair (timing_begin .. timing_end): a failure closes the pair before it returns,
// so that smmc_engine_kernel_ms never reads a start event without its stop.
#define SMMC_HIP_TIMED(e, call)                                                                \
  do {                                                                                         \
    hipError_t err__ = (call);                                                                 \
    if (err__ != hipSuccess) {                                                                 \
      (void)timing_end(e);                                                                     \
      return fail(SMMC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__),     \
                  __FILE__, __LINE__);                                                         \
    }                                                                                          \
  } while (0)

// SMMC_VERBOSE=2: where the host time of the set-up paths goes (engine creation, the first
// simulate_to_host of a process): one stderr line per phase, cumulative milliseconds.
struct PhaseLog {
  const char *what;
  bool on;
  std::chrono::steady_clock::time_point t0;
  explicit PhaseLog(const char *w) : what(w), t0(std::chrono::steady_clock::now()) {
    const char *env = std::getenv("SMMC_VERBOSE");
    on = env && env[0] >= '2' && env[0] <= '9';
  }
  void mark(const char *phase) const {
    if (on)
      std::fprintf(stderr, "smmc: %s: +%.2f ms %s\n", what,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), phase);
  }
};

// Makes `device` current for the scope and restores the caller's device after.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) ok = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

// Workgroups launched per CU.  About 6 are resident (SGPR-limited); the rest queue and
// are handed out as others finish, which evens out CU/XCD speed differences: measured
// 1e8 x 360 table paths: 13.1 ms at 6 per CU, 12.6 at 8, 11.7 at 16, 11.1 at 64..256.
constexpr uint32_t kBlocksPerCU = 64;
// simulate_to_host: 16 MiB of floats per chunk.  1e8 x 360 table paths into pinned memory, warm (tools/bench_host_chunks.py,
// profiles/r03/host_chunks.jsonl): 2^24-path chunks 9.1 ms, 2^23 8.2, 2^22 8.0, 2^21 8.4 polled / 11.6 not -- shorter chunks
// start the first copy earlier and leave a shorter last one; below 2^22 the host's enqueue rate shows.
constexpr uint64_t kHostChunkPaths = 1ull << 22;
constexpr uint64_t kReduceChunkValues = 1ull << 24;  // reduce_mean_host: 64 MiB host-to-device pieces
constexpr uint64_t kProgressChunkMin = 1ull << 16;   // ... and at least this many when progress is polled

}  // namespace

struct smmc_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t copy_stream = nullptr;  // lazily created, simulate_to_host only
  uint32_t compute_units = 0;
  uint32_t max_grid = 0;
  uint32_t keepdata_blocks_per_cu = 0;  // 0: as many keepdata workgroups as are resident
  size_t max_lds = 0;

  float *d_table = nullptr;  // 100.0f + r
  uint32_t table_len = 0;
  float table_min_a = 0.f, table_max_a = 0.f;
  bool table_finite = false;

  float *d_bm_tables = nullptr;  // Box-Muller radius + trig tables (Gaussian mode)
  smmc::BlockPartial *d_partials = nullptr;  // max_grid entries
  unsigned long long *d_clock = nullptr;     // timing: [shader clocks, 100 MHz ticks] summed over paths_kernel's workgroups

  // simulate_to_host staging
  float *d_stage[2] = {nullptr, nullptr};
  uint64_t stage_paths = 0;
  float *d_stage_cs[2] = {nullptr, nullptr};  // chunk means then variances
  uint64_t stage_cs = 0;
  void *d_stage_stats = nullptr;
  size_t stage_stats_bytes = 0;
  hipEvent_t ev_compute[2] = {nullptr, nullptr};
  hipEvent_t ev_copy[2] = {nullptr, nullptr};
  hipEvent_t ev_order = nullptr;  // smmc_engine_set_stream / wait_stream / release_to_stream

  // order statistics workspace
  smmc::SelectState *d_select = nullptr;
  unsigned long long *d_radix_hist = nullptr;  // kMaxRanks x 2048; zero between passes and calls (radix_pick_kernel
                                               // clears what a pass counted); radix_dirty: an enqueue failed in between
  bool radix_dirty = false;
  smmc::SelectState *h_select = nullptr;       // page-locked: the call's ranks go up from here without a sync
  float *d_select_out = nullptr;               // kMaxRanks
  void *d_scratch_stats = nullptr;             // one packed record with SMMC_MAX_BINS buckets
  unsigned long long *d_work_counter = nullptr;  // the comb keepdata kernel's chunk queue
  // Bucket counts of the launch in flight: paths_kernel adds into copy 0, values_stats into kHistSpread copies;
  // finalize_kernel folds them into the caller's record and zeroes them again, so the array is ZERO between
  // launches and neither it nor the record is memset per call.  hist_dirty: an enqueue failed between the kernel
  // and its finalize -- the next user clears the whole array first (hist_acc_ready).
  unsigned long long *d_hist_spread = nullptr;
  bool hist_dirty = false;

  // SMMC_FLAG_STREAM_REF (smmc_ref_kernels.hip)
  float *d_ref_final = nullptr;     // final values of a launch that asked for none (statistics are formed from them)
  uint64_t ref_final_cap = 0;
  uint32_t *d_ref_redo = nullptr;   // [0]: count, [4 ..): paths the windowed kernel left to the generic one
  uint64_t ref_redo_cap = 0;
  uint32_t *d_ref_ws = nullptr;     // the generic kernel's generator states
  uint32_t ref_ws_grid = 0;
  int ref_kernel = 0;               // SMMC_REF_KERNEL: 0 auto (also "tree", read by the launcher), 1 windowed (where it can), 2 generic
  // SMMC_REF_GENERIC_BLOCKS_PER_CU.  2e7 x 1000 paths (profiles/r03/ref_generic_sweep.txt): 1 per CU 85.8 ms, 2 60.0,
  // 4 48.3, 8 35.0 -- the chains that now supply the seed words want the occupancy; the 1.3 GB of generator
  // states no longer fit the Infinity Cache, but a path moves 41 % fewer bytes than when its seed words were stored
  uint32_t ref_generic_per_cu = 8;

  bool timing = false;
  std::vector<hipEvent_t> ev_pool;  // pairs: start, stop
  size_t ev_used = 0;

  uint64_t host_chunk_paths = kHostChunkPaths;  // SMMC_HOST_CHUNK_PATHS
  int pin_policy = smmc::kPinWhole;             // SMMC_PIN_HOST (smmc_host.h): never, whole buffer (default), chunk by chunk
  uint64_t pin_min_bytes = smmc::kPinMinBytes;
  smmc_progress_fn progress_fn = nullptr;
  void *progress_user = nullptr;
};

namespace {

int check_sim(const smmc_engine *e, const smmc_sim *s) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (!s) return fail(SMMC_ERR_INVALID, "sim is NULL");
  if (s->struct_size != sizeof(smmc_sim))
    return fail(SMMC_ERR_INVALID, "smmc_sim.struct_size is %u, this library expects %zu", s->struct_size,
                sizeof(smmc_sim));
  if (s->mode != SMMC_MODE_TABLE && s->mode != SMMC_MODE_GAUSSIAN)
    return fail(SMMC_ERR_INVALID, "unknown mode %d", s->mode);
  if (s->mode == SMMC_MODE_TABLE && e->table_len == 0)
    return fail(SMMC_ERR_INVALID, "table mode needs smmc_engine_set_table first");
  if (s->n_bins > SMMC_MAX_BINS)
    return fail(SMMC_ERR_INVALID, "n_bins %u exceeds SMMC_MAX_BINS %d", s->n_bins, SMMC_MAX_BINS);
  if (s->n_bins && !(s->hist_lo < s->hist_hi))
    return fail(SMMC_ERR_INVALID, "histogram range must satisfy lo < hi (got %g, %g)", s->hist_lo, s->hist_hi);
  if (s->n_bins && (!std::isfinite(s->hist_lo) || !std::isfinite(s->hist_hi)))
    return fail(SMMC_ERR_INVALID, "histogram range must be finite");
  if (s->flags & SMMC_FLAG_STREAM_REF) {
    if (s->mode != SMMC_MODE_TABLE)
      return fail(SMMC_ERR_INVALID, "SMMC_FLAG_STREAM_REF is the reference's table-draw engine (src/simulations.cpp:240-252): table mode only");
    if (s->flags & SMMC_FLAG_STREAM_V2) return fail(SMMC_ERR_INVALID, "SMMC_FLAG_STREAM_REF and SMMC_FLAG_STREAM_V2 exclude each other");
  }
  if (s->n_paths > (1ull << 62)) return fail(SMMC_ERR_INVALID, "n_paths too large");
  if (s->n_periods >= (1u << 31)) return fail(SMMC_ERR_INVALID, "n_periods too large");
  return SMMC_OK;
}

// The two-instruction divide (div100, smmc_kernels.hip) is exact for every finite |x| >= 2^-114 (and
// x = +0), x being the product total * a; the host keeps to |x| >= 2^-90, where its low product is a
// normal number too.  Bounds on a = 100 + r for one period: false if there are none.
bool multiplier_bounds(const smmc_engine *e, const smmc_sim *s, double *lo_a, double *hi_a) {
  if (s->mode == SMMC_MODE_TABLE) {
    if (!e->table_finite) return false;
    *lo_a = e->table_min_a;
    *hi_a = e->table_max_a;
  } else {
    if (!std::isfinite(s->gauss_mean) || !std::isfinite(s->gauss_std)) return false;
    // |z| <= sqrt(-2 ln 2^-33) * (1 + eps) < 6.8; 7 leaves margin for the roundings
    const double spread = 7.0 * std::fabs(static_cast<double>(s->gauss_std));
    *lo_a = 100.0 + s->gauss_mean - spread - 1e-3;
    *hi_a = 100.0 + s->gauss_mean + spread + 1e-3;
  }
  return *lo_a > 0.0 && std::isfinite(*hi_a);
}

// Which divide a launch may use (SMMC_DIV_*).  FAST: no product of any path can leave
// [2^-89, 2^127) or change sign, proven from the bounds, the capital and the number of periods
// (one bit of margin on each side for the roundings along the way).  CHECKED (paths_kernel only):
// not provable -- a table with one +42 % month fails for 360 periods -- but a path inside
// [*chk_lo, *chk_hi] at a Philox-block boundary cannot leave the domain during the next 8 periods,
// so the kernel tests that window once per block and redoes the rare path that leaves it with the
// IEEE divide.  EXACT otherwise, or on request.
constexpr double kFastDivLog2Min = -89.0;  // products stay above 2^-89: one bit above the 2^-90 the divide is used from
int divide_kind(const smmc_engine *e, const smmc_sim *s, bool allow_checked, float *chk_lo, float *chk_hi) {
  *chk_lo = 0.0f;
  *chk_hi = 0.0f;
  if (s->flags & SMMC_FLAG_EXACT_DIV) return SMMC_DIV_EXACT;
  const double cap = s->initial_capital;
  if (!(cap > 0.0) || !std::isfinite(cap)) return SMMC_DIV_EXACT;
  double lo_a, hi_a;
  if (!multiplier_bounds(e, s, &lo_a, &hi_a)) return SMMC_DIV_EXACT;
  const double p = static_cast<double>(s->n_periods);
  const double grow = std::max(0.0, std::log2(hi_a / 100.0)), shrink = std::max(0.0, -std::log2(lo_a / 100.0));
  const double top = std::log2(hi_a), bottom = std::min(0.0, std::log2(lo_a));
  // total after k periods lies in cap * [(lo_a/100)^k, (hi_a/100)^k]; the next product is that times a
  if (std::log2(cap) + p * grow + top + 1.0 < 127.0 && std::log2(cap) - p * shrink + bottom - 1.0 > kFastDivLog2Min)
    return SMMC_DIV_FAST;
  if (!allow_checked) return SMMC_DIV_EXACT;
  // a block is at most 8 periods: 7 steps to its last total, then one more product
  const double hi_w = 127.0 - 1.0 - 7.0 * grow - top, lo_w = kFastDivLog2Min + 1.0 + 7.0 * shrink - bottom;
  if (!(lo_w + 2.0 < std::log2(cap) && std::log2(cap) < hi_w - 2.0)) return SMMC_DIV_EXACT;
  *chk_lo = static_cast<float>(std::exp2(lo_w));
  *chk_hi = static_cast<float>(std::exp2(hi_w));
  return SMMC_DIV_CHECKED;
}

smmc::KernelArgs make_args(const smmc_engine *e, const smmc_sim *s) {
  smmc::KernelArgs a;
  std::memset(&a, 0, sizeof a);
  a.mode = s->mode;
  a.table_a = s->mode == SMMC_MODE_TABLE ? e->d_table : nullptr;
  a.table_len = s->mode == SMMC_MODE_TABLE ? e->table_len : 0u;
  a.stream = (s->flags & SMMC_FLAG_STREAM_V2) ? 2 : 3;
  // d_bm_tables holds v2's radius + trig tables, then v3's
  a.bm_tables = a.stream == 2 ? e->d_bm_tables : e->d_bm_tables + smmc::bm_tables_bytes(2) / sizeof(float);
  a.key0 = static_cast<uint32_t>(s->seed);
  a.key1 = static_cast<uint32_t>(s->seed >> 32);
  a.first_path = s->first_path;
  a.n_paths = s->n_paths;
  a.n_periods = s->n_periods;
  a.initial_capital = s->initial_capital;
  a.gauss_mean = s->gauss_mean;
  a.gauss_std = s->gauss_std;
  a.gauss_shift100 = 100.0f + s->gauss_mean;
  a.n_bins = s->n_bins;
  a.hist_lo = s->hist_lo;
  a.hist_hi = s->hist_hi;
  a.hist_inv = s->n_bins ? static_cast<double>(s->n_bins) /
                               (static_cast<double>(s->hist_hi) - static_cast<double>(s->hist_lo))
                         : 0.0;
  a.below_threshold = s->below_threshold;
  return a;
}

int timing_begin(smmc_engine *e) {
  if (!e->timing) return SMMC_OK;
  if (e->ev_used + 2 > e->ev_pool.size()) {
    for (int i = 0; i < 2; ++i) {
      hipEvent_t ev;
      SMMC_HIP(hipEventCreate(&ev));
      e->ev_pool.push_back(ev);
    }
  }
  SMMC_HIP(hipEventRecord(e->ev_pool[e->ev_used], e->stream));
  return SMMC_OK;
}
int timing_end(smmc_engine *e) {
  if (!e->timing) return SMMC_OK;
  SMMC_HIP(hipEventRecord(e->ev_pool[e->ev_used + 1], e->stream));
  e->ev_used += 2;
  return SMMC_OK;
}

// One pass over n device floats -> packed statistics record (smmc_engine_values_stats after its argument
// checks; also the statistics of a SMMC_FLAG_STREAM_REF launch).  Device must be current.
// The engine's bucket accumulator, allocated and cleared on first use and cleared again after a failed enqueue.
int hist_acc_ready(smmc_engine *e) {
  const size_t bytes = sizeof(unsigned long long) * smmc::kHistSpread * SMMC_MAX_BINS;
  if (!e->d_hist_spread) {
    SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_hist_spread), bytes));
    e->hist_dirty = true;
  }
  if (e->hist_dirty) {
    SMMC_HIP(hipMemsetAsync(e->d_hist_spread, 0, bytes, e->stream));
    e->hist_dirty = false;
  }
  return SMMC_OK;
}

int values_stats_enqueue(smmc_engine *e, const float *d_values, uint64_t n, float below_threshold, uint32_t n_bins,
                         float hist_lo, float hist_hi, void *d_stats, bool timed) {
  smmc::ValuesArgs a;
  std::memset(&a, 0, sizeof a);
  a.values = d_values;
  a.n = n;
  a.below_threshold = below_threshold;
  a.n_bins = n_bins;
  a.hist_copies = smmc::values_hist_copies(n_bins);
  a.hist_lo = hist_lo;
  a.hist_hi = hist_hi;
  a.hist_inv = n_bins ? static_cast<double>(n_bins) / (static_cast<double>(hist_hi) - static_cast<double>(hist_lo)) : 0.0;
  a.partials = e->d_partials;
  a.d_hist = reinterpret_cast<unsigned long long *>(static_cast<char *>(d_stats) + sizeof(smmc_stats));
  if (n_bins) {  // spread copies of the bucket array (see ValuesArgs), zero now and zero again after finalize_kernel
    const int rc = hist_acc_ready(e);
    if (rc) return rc;
    a.hist_spread = e->d_hist_spread;
    a.spread = smmc::kHistSpread;
  }
  // 16 bytes per lane per iteration, 1024-thread workgroups, four per CU: two are resident (32 waves
  // per CU), the other two queue and even out the CUs' speeds.  1e8 / 1e9 values with the 100-bucket
  // histogram: 1 per CU 0.102 / 0.758 ms, 2 per CU 0.083 / 0.699, 4 per CU 0.080 / 0.676
  const uint64_t want = (n / 4 + 1023) / 1024;
  uint32_t per_cu = 4;
  if (const char *env = std::getenv("SMMC_STATS_BLOCKS_PER_CU")) {  // tuning knob
    const long v = std::strtol(env, nullptr, 10);
    if (v >= 1 && v <= 16) per_cu = static_cast<uint32_t>(v);
  }
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint64_t>(want, 1), std::min(e->compute_units * per_cu, e->max_grid)));
  e->hist_dirty = n_bins != 0;  // until the finalize that clears the accumulator is in the queue
  if (n) {
    int rc = timed ? timing_begin(e) : SMMC_OK;
    if (rc) return rc;
    const hipError_t err = smmc::launch_values_stats(a, grid, e->stream);
    if (err != hipSuccess) {
      if (timed) (void)timing_end(e);
      return fail(SMMC_ERR_HIP, "launch_values_stats failed: %s", hipGetErrorString(err));
    }
    rc = timed ? timing_end(e) : SMMC_OK;
    if (rc) return rc;
  }
  SMMC_HIP(smmc::launch_finalize(e->d_partials, n ? grid : 0u, static_cast<smmc_stats *>(d_stats), n_bins, e->stream,
                                 a.hist_spread, a.spread));
  e->hist_dirty = false;
  return SMMC_OK;
}

// SMMC_FLAG_STREAM_REF: the reference's own stream (smmc_ref_kernels.hip).  Launches of at most 2^27 paths:
// a state-free kernel where it can generate a path's outputs (launch_ref_windowed: the windowed kernel up to
// 454 periods, the tree kernel up to ref_windowed_max_outputs() = 1816; SMMC_REF_KERNEL=tree: the tree kernel
// for both), then a small generic launch over the paths it left (a rejected generator output -- 1e-4 of the
// paths with the 1127-entry table at 360 periods -- or a path that left the checked divide's window); the
// generic kernel for everything when n_periods is larger.
// Statistics and chunk outputs are second passes over the final values.  Device must be current.
constexpr uint64_t kRefLaunchPaths = 1ull << 27;
constexpr uint32_t kRefRedoGrid = 64;

int enqueue_ref_simulation(smmc_engine *e, const smmc_sim *s, float *d_final, float *d_chunk_mean, float *d_chunk_var,
                           void *d_stats, float *d_traj = nullptr) {
  const uint64_t n = s->n_paths;
  float *fin = d_final;
  if (!fin && n) {
    if (e->ref_final_cap < n) {
      SMMC_HIP(hipStreamSynchronize(e->stream));
      if (e->d_ref_final) SMMC_HIP(hipFree(e->d_ref_final));
      e->d_ref_final = nullptr;
      e->ref_final_cap = 0;
      SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_ref_final), sizeof(float) * n));
      e->ref_final_cap = n;
    }
    fin = e->d_ref_final;
  }
  const uint32_t T = e->table_len;
  const uint32_t P = s->n_periods;
  const bool windowed = e->ref_kernel != 2 && P <= smmc::ref_windowed_max_outputs();
  const uint32_t full_grid = e->compute_units * e->ref_generic_per_cu;
  if (n) {
    const uint64_t seg = std::min<uint64_t>(n, kRefLaunchPaths);
    if (windowed && e->ref_redo_cap < seg) {
      SMMC_HIP(hipStreamSynchronize(e->stream));
      if (e->d_ref_redo) SMMC_HIP(hipFree(e->d_ref_redo));
      e->d_ref_redo = nullptr;
      e->ref_redo_cap = 0;
      SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_ref_redo), sizeof(uint32_t) * (seg + 4)));
      e->ref_redo_cap = seg;
    }
    const uint32_t ws_grid = windowed ? kRefRedoGrid : full_grid;
    if (e->ref_ws_grid < ws_grid) {
      SMMC_HIP(hipStreamSynchronize(e->stream));
      if (e->d_ref_ws) SMMC_HIP(hipFree(e->d_ref_ws));
      e->d_ref_ws = nullptr;
      e->ref_ws_grid = 0;
      SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_ref_ws), smmc::ref_workspace_bytes(ws_grid)));
      e->ref_ws_grid = ws_grid;
    }
    if (smmc::ref_windowed_lds_bytes(T, d_traj != nullptr) + 2048 > e->max_lds)
      return fail(SMMC_ERR_INVALID, "the table needs %zu bytes of LDS, device allows %zu",
                  smmc::ref_windowed_lds_bytes(T, d_traj != nullptr), e->max_lds);
    smmc::RefArgs a;
    std::memset(&a, 0, sizeof a);
    a.table_a = e->d_table;
    a.table_len = T;
    a.reject_below = static_cast<uint32_t>((0x100000000ull - T) % T);  // uniform_int_dist.h:258: -range % range
    a.n_periods = P;
    a.initial_capital = s->initial_capital;
    a.workspace = e->d_ref_ws;
    int div = divide_kind(e, s, true, &a.chk_lo, &a.chk_hi);
    int rc = timing_begin(e);
    if (rc) return rc;
    for (uint64_t first = 0; first < n; first += seg) {
      const uint64_t count = std::min<uint64_t>(seg, n - first);
      a.seed0 = static_cast<uint32_t>(s->seed + s->first_path + first);
      a.n_paths = static_cast<uint32_t>(count);
      a.d_final = fin + first;
      a.d_traj = d_traj ? d_traj + first * (static_cast<uint64_t>(P) + 1) : nullptr;
      // trajectories: consecutive rows per lane.  More rows are fewer junction lines and fewer, longer work units:
      // the largest of 8 / 4 / 2 / 1 that still makes one and a half rounds of the resident workgroups (four per CU:
      // the tiles' LDS).  Measured (profiles/r04/bench_ref_traj.jsonl): 4e6 x 361 values K = 8 / 4 / 2 / 1 1.78 / 1.98 /
      // 2.11 / 2.18 ms (1953 units at K = 8); 1.5e6 x 1001 3.21 / 2.99 / 2.75 / 2.82 ms (733 units at K = 8: the chip
      // is not filled).  SMMC_REF_TRAJ_ROWS forces one; results do not depend on it.
      a.traj_rows = 8;
      if (d_traj) {
        const uint64_t n_super = (count + 2047) / 2048, want = 6ull * e->compute_units;
        while (a.traj_rows > 1 && n_super * (8u / a.traj_rows) < want) a.traj_rows /= 2;
        if (const char *env = std::getenv("SMMC_REF_TRAJ_ROWS")) {  // tuning knob
          const long v = std::strtol(env, nullptr, 10);
          if (v == 8 || v == 4 || v == 2 || v == 1) a.traj_rows = static_cast<uint32_t>(v);
        }
      }
      const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>((count + smmc::kBlock - 1) / smmc::kBlock, e->max_grid));
      if (windowed) {
        a.redo_count = e->d_ref_redo;
        a.redo_list = e->d_ref_redo + 4;
        SMMC_HIP_TIMED(e, hipMemsetAsync(e->d_ref_redo, 0, sizeof(uint32_t), e->stream));
        SMMC_HIP_TIMED(e, smmc::launch_ref_windowed(a, div, grid, e->stream));
        SMMC_HIP_TIMED(e, smmc::launch_ref_generic(a, true, kRefRedoGrid, e->stream));
      } else {
        a.redo_count = nullptr;
        a.redo_list = nullptr;
        SMMC_HIP_TIMED(e, smmc::launch_ref_generic(a, div != SMMC_DIV_FAST, std::min(grid, full_grid), e->stream));
      }
    }
    rc = timing_end(e);
    if (rc) return rc;
  }
  if (d_chunk_mean || d_chunk_var) {
    const uint64_t n_chunks = (n + smmc::kBlock - 1) / smmc::kBlock;
    if (n_chunks)
      SMMC_HIP(smmc::launch_chunk_stats(fin, n, d_chunk_mean, d_chunk_var,
                                        static_cast<uint32_t>(std::min<uint64_t>(n_chunks, e->max_grid)), e->stream));
  }
  if (d_stats)
    return values_stats_enqueue(e, fin, n, s->below_threshold, s->n_bins, s->hist_lo, s->hist_hi, d_stats, false);
  return SMMC_OK;
}

// Enqueue: paths kernel -> finalize (two launches: the record is written whole by finalize_kernel, the bucket
// accumulator is left zero by it).  Device must be current.
int enqueue_simulation(smmc_engine *e, const smmc_sim *s, float *d_final, float *d_chunk_mean,
                       float *d_chunk_var, void *d_stats) {
  if (s->flags & SMMC_FLAG_STREAM_REF) return enqueue_ref_simulation(e, s, d_final, d_chunk_mean, d_chunk_var, d_stats);
  smmc::KernelArgs a = make_args(e, s);
  a.d_final = d_final;
  a.d_chunk_mean = d_chunk_mean;
  a.d_chunk_var = d_chunk_var;
  const uint64_t n_chunks = (s->n_paths + smmc::kBlock - 1) / smmc::kBlock;
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>(n_chunks, e->max_grid));
  if (d_stats) {  // no memset: finalize_kernel writes the whole record and leaves the accumulator zero again
    a.partials = e->d_partials;
    if (s->n_bins) {
      const int rc = hist_acc_ready(e);
      if (rc) return rc;
      a.d_hist = e->d_hist_spread;  // copy 0
      e->hist_dirty = true;
    }
  }
  a.clock_probe = e->timing ? e->d_clock : nullptr;
  if (grid > 0) {
    const size_t lds = smmc::paths_lds_bytes(a.table_len, d_stats ? s->n_bins : 0u, a.stream);
    if (lds + 2048 > e->max_lds)
      return fail(SMMC_ERR_INVALID, "table + histogram need %zu bytes of LDS, device allows %zu", lds, e->max_lds);
    const int div = divide_kind(e, s, true, &a.chk_lo, &a.chk_hi);
    int rc = timing_begin(e);
    if (rc) return rc;
    const hipError_t err = smmc::launch_paths(a, div, grid, lds, e->stream);
    if (err != hipSuccess) {
      (void)timing_end(e);
      return fail(SMMC_ERR_HIP, "launch_paths failed: %s", hipGetErrorString(err));
    }
    rc = timing_end(e);
    if (rc) return rc;
  }
  if (d_stats) {
    SMMC_HIP(smmc::launch_finalize(e->d_partials, grid, static_cast<smmc_stats *>(d_stats), s->n_bins, e->stream,
                                   s->n_bins ? e->d_hist_spread : nullptr, s->n_bins ? 1u : 0u));
    e->hist_dirty = false;
  }
  return SMMC_OK;
}

}  // namespace

extern "C" {

int smmc_abi_version(void) { return SMMC_ABI_VERSION; }
const char *smmc_last_error(void) { return g_err; }
// for the library's other translation units (smmc_group.cpp): the calling thread's error text.  Not part of the
// ABI: hidden, so that libsmmc_hip.so exports exactly what include/smmc.h declares besides the C++ drop-in layer.
__attribute__((visibility("hidden"))) int smmc_set_error_(int code, const char *message) {
  return fail(code, "%s", message ? message : "");
}

float smmc_update_fund(float fund_value, float period_return) {
  // reference src/simulations.cpp:14-16; this TU is built with -ffp-contract=off
  const float a = 100.0f + period_return;
  const float m = fund_value * a;
  return m / 100.0f;
}

void smmc_many_updates(const float *returns, float *totals, uint32_t n_periods) {
  // reference src/simulations.cpp:18-22
  float t = totals[0];
  for (uint32_t i = 0; i < n_periods; ++i) {
    t = smmc_update_fund(t, returns[i]);
    totals[i + 1] = t;
  }
}

int smmc_device_count(int *count) {
  if (!count) return fail(SMMC_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return SMMC_OK;
}

int smmc_engine_create(int device, void *stream, smmc_engine **out) {
  if (!out) return fail(SMMC_ERR_INVALID, "out is NULL");
  *out = nullptr;
  const PhaseLog phase("engine_create");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    (void)hipGetLastError();
    return fail(SMMC_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
  }
  if (device < 0 || device >= n) return fail(SMMC_ERR_INVALID, "device %d out of range [0, %d)", device, n);
  DeviceGuard guard(device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", device);
  phase.mark("runtime up, device current");
  hipDeviceProp_t prop;
  SMMC_HIP(hipGetDeviceProperties(&prop, device));
  phase.mark("device properties");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(SMMC_ERR_NO_DEVICE, "device %d is %s; this library only carries gfx950 code", device, prop.gcnArchName);
  smmc_engine *e = new (std::nothrow) smmc_engine();
  if (!e) return fail(SMMC_ERR_NOMEM, "out of host memory");
  e->device = device;
  e->compute_units = static_cast<uint32_t>(prop.multiProcessorCount);
  uint32_t blocks_per_cu = kBlocksPerCU;
  if (const char *env = std::getenv("SMMC_BLOCKS_PER_CU")) {  // tuning knob, results do not depend on it
    const long v = std::strtol(env, nullptr, 10);
    if (v >= 1 && v <= 4096) blocks_per_cu = static_cast<uint32_t>(v);
  }
  e->max_grid = e->compute_units * blocks_per_cu;
  if (const char *env = std::getenv("SMMC_KEEPDATA_BLOCKS_PER_CU")) {
    const long v = std::strtol(env, nullptr, 10);
    if (v >= 1 && v <= 4096) e->keepdata_blocks_per_cu = static_cast<uint32_t>(v);
  }
  if (const char *env = std::getenv("SMMC_HOST_CHUNK_PATHS")) {  // tuning/test knob, results do not depend on it
    const long long v = std::strtoll(env, nullptr, 10);
    if (v >= 1024) e->host_chunk_paths = static_cast<uint64_t>(v) / 1024 * 1024;
    else if (v >= smmc::kBlock) e->host_chunk_paths = static_cast<uint64_t>(v) / smmc::kBlock * smmc::kBlock;  // tests
  }
  e->pin_policy = smmc::pin_policy_from_env();  // SMMC_PIN_HOST: see smmc_engine_simulate_to_host, smmc_host.h
  if (const char *env = std::getenv("SMMC_REF_KERNEL")) {  // test knob, results do not depend on it
    e->ref_kernel = !std::strcmp(env, "windowed") ? 1 : !std::strcmp(env, "generic") ? 2 : 0;
  }
  if (const char *env = std::getenv("SMMC_REF_GENERIC_BLOCKS_PER_CU")) {  // tuning knob
    const long v = std::strtol(env, nullptr, 10);
    if (v >= 1 && v <= 16) e->ref_generic_per_cu = static_cast<uint32_t>(v);
  }
  // dynamic LDS a launch may ask for: the kernels opt in above the 64 KiB default (CDNA4: 160 KiB per CU)
  e->max_lds = std::max<size_t>(prop.sharedMemPerBlock, 128u * 1024u);
  // The first stream of a process costs ~20 ms (its hardware queue), the first host-to-device copy ~17 ms,
  // the side stream of the host pipeline 5-9 ms (SMMC_VERBOSE=2 prints the phases; profiles/r03/cold_start.txt).
  // Bringing the three up from three host threads at once was tried: the runtime serialises them
  // (stream 20 -> 33 ms, everything done after 47-59 ms instead of 46-50), so they run one after the other.
  static_assert(sizeof(smmc_bm_radius) + sizeof(smmc_bm_trig) == (1056 * 4 + 256 * 2) * 4, "v2 table layout");
  static_assert(sizeof(smmc_bm3_radius) + sizeof(smmc_bm3_trig) == (512 * 4 + 2048 * 2) * 4, "v3 table layout");
  static_assert(smmc::kBm3SubBits == SMMC_BM3_SUB_BITS && smmc::kBm3TrigBits == SMMC_BM3_TRIG_BITS &&
                    smmc::kBm3AngleBits == SMMC_BM3_ANGLE_BITS && smmc::kBm3AngleK == SMMC_BM3_ANGLE_K &&
                    smmc::kBm3AngleC == SMMC_BM3_ANGLE_C,
                "the kernels' v3 constants differ from the generated tables");
  if (smmc::bm_tables_bytes(2) != sizeof(smmc_bm_radius) + sizeof(smmc_bm_trig) ||
      smmc::bm_tables_bytes(3) != sizeof(smmc_bm3_radius) + sizeof(smmc_bm3_trig)) {
    delete e;
    return fail(SMMC_ERR_INVALID, "Box-Muller table size mismatch between host and kernels");
  }
  hipError_t err_stream = hipSuccess, err_side = hipSuccess, err_tables = hipSuccess;
  size_t static_lds = 0;
  auto make_stream = [&]() {
    if (stream != SMMC_STREAM_NEW) {
      e->stream = static_cast<hipStream_t>(stream);  // NULL = the default stream
      return;
    }
    err_stream = hipSetDevice(device);
    if (err_stream == hipSuccess) err_stream = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    e->own_stream = err_stream == hipSuccess;
    phase.mark("stream");
  };
  auto make_side = [&]() {  // what simulate_to_host and reduce_mean_host would otherwise create on first use
    err_side = hipSetDevice(device);
    if (err_side == hipSuccess) err_side = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && err_side == hipSuccess; ++i) {
      err_side = hipEventCreateWithFlags(&e->ev_compute[i], hipEventDisableTiming);
      if (err_side == hipSuccess) err_side = hipEventCreateWithFlags(&e->ev_copy[i], hipEventDisableTiming);
    }
    phase.mark("side stream and events");
  };
  auto make_tables = [&]() {
    err_tables = hipSetDevice(device);
    if (err_tables == hipSuccess)
      err_tables = hipMalloc(reinterpret_cast<void **>(&e->d_partials), sizeof(smmc::BlockPartial) * e->max_grid);
    // the Gaussian kernels read counter stream v3's tables at absolute LDS addresses from 0
    if (err_tables == hipSuccess) err_tables = smmc::static_lds_bytes(&static_lds);
    phase.mark("kernel attributes (code object loaded)");
    // Box-Muller tables, as the kernels stage them: radius cubics then (cos, sin) pairs, counter stream
    // v2's set first, then v3's
    if (err_tables == hipSuccess)
      err_tables = hipMalloc(reinterpret_cast<void **>(&e->d_bm_tables), smmc::bm_tables_bytes(2) + smmc::bm_tables_bytes(3));
    char *dst = reinterpret_cast<char *>(e->d_bm_tables);
    const void *parts[4] = {smmc_bm_radius, smmc_bm_trig, smmc_bm3_radius, smmc_bm3_trig};
    const size_t sizes[4] = {sizeof(smmc_bm_radius), sizeof(smmc_bm_trig), sizeof(smmc_bm3_radius), sizeof(smmc_bm3_trig)};
    for (int i = 0; i < 4 && err_tables == hipSuccess; ++i) {
      err_tables = hipMemcpy(dst, parts[i], sizes[i], hipMemcpyHostToDevice);
      dst += sizes[i];
    }
    phase.mark("Box-Muller tables uploaded");
  };
  make_stream();
  if (err_stream == hipSuccess) make_tables();
  if (err_stream == hipSuccess && err_tables == hipSuccess) make_side();
  if (err_stream != hipSuccess || err_side != hipSuccess || err_tables != hipSuccess) {
    const hipError_t err = err_stream != hipSuccess ? err_stream : err_side != hipSuccess ? err_side : err_tables;
    if (!e->own_stream && stream == SMMC_STREAM_NEW) e->stream = nullptr;
    smmc_engine_destroy(e);
    return fail(SMMC_ERR_HIP, "bringing up the engine on device %d failed: %s", device, hipGetErrorString(err));
  }
  if (static_lds != 0) {
    smmc_engine_destroy(e);
    return fail(SMMC_ERR_INVALID, "a Gaussian kernel has %zu bytes of static LDS; the draw tables must start at LDS address 0", static_lds);
  }
  phase.mark("engine up");
  *out = e;
  return SMMC_OK;
}

void smmc_engine_destroy(smmc_engine *e) {
  if (!e) return;
  DeviceGuard guard(e->device);
  (void)hipStreamSynchronize(e->stream);
  if (e->copy_stream) {
    (void)hipStreamSynchronize(e->copy_stream);
    (void)hipStreamDestroy(e->copy_stream);
  }
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  if (e->ev_order) (void)hipEventDestroy(e->ev_order);
  for (int i = 0; i < 2; ++i) {
    if (e->ev_compute[i]) (void)hipEventDestroy(e->ev_compute[i]);
    if (e->ev_copy[i]) (void)hipEventDestroy(e->ev_copy[i]);
    if (e->d_stage[i]) (void)hipFree(e->d_stage[i]);
    if (e->d_stage_cs[i]) (void)hipFree(e->d_stage_cs[i]);
  }
  if (e->d_stage_stats) (void)hipFree(e->d_stage_stats);
  if (e->d_table) (void)hipFree(e->d_table);
  if (e->d_bm_tables) (void)hipFree(e->d_bm_tables);
  if (e->d_select) (void)hipFree(e->d_select);
  if (e->d_radix_hist) (void)hipFree(e->d_radix_hist);
  if (e->h_select) (void)hipHostFree(e->h_select);
  if (e->d_select_out) (void)hipFree(e->d_select_out);
  if (e->d_scratch_stats) (void)hipFree(e->d_scratch_stats);
  if (e->d_work_counter) (void)hipFree(e->d_work_counter);
  if (e->d_hist_spread) (void)hipFree(e->d_hist_spread);
  if (e->d_ref_final) (void)hipFree(e->d_ref_final);
  if (e->d_ref_redo) (void)hipFree(e->d_ref_redo);
  if (e->d_ref_ws) (void)hipFree(e->d_ref_ws);
  if (e->d_partials) (void)hipFree(e->d_partials);
  if (e->d_clock) (void)hipFree(e->d_clock);
  if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

int smmc_engine_set_table(smmc_engine *e, const float *returns_percent, uint32_t n) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (!returns_percent || n == 0) return fail(SMMC_ERR_INVALID, "empty returns table");
  if (n > SMMC_MAX_TABLE) return fail(SMMC_ERR_INVALID, "table of %u entries exceeds SMMC_MAX_TABLE %d", n, SMMC_MAX_TABLE);
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  // a = 100.0f + r: the first of update_fund's three roundings, hoisted out of the loop
  std::vector<float> a(n);
  float lo = std::numeric_limits<float>::infinity(), hi = -lo;
  bool finite = true;
  for (uint32_t i = 0; i < n; ++i) {
    a[i] = 100.0f + returns_percent[i];
    finite = finite && std::isfinite(a[i]);
    lo = std::min(lo, a[i]);
    hi = std::max(hi, a[i]);
  }
  if (e->table_len != n) {
    // the previous table may still be read by enqueued kernels
    SMMC_HIP(hipStreamSynchronize(e->stream));
    if (e->d_table) SMMC_HIP(hipFree(e->d_table));
    e->d_table = nullptr;
    e->table_len = 0;
    SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_table), sizeof(float) * n));
  }
  SMMC_HIP(hipMemcpyAsync(e->d_table, a.data(), sizeof(float) * n, hipMemcpyHostToDevice, e->stream));
  SMMC_HIP(hipStreamSynchronize(e->stream));  // `a` is a local
  e->table_len = n;
  e->table_min_a = lo;
  e->table_max_a = hi;
  e->table_finite = finite;
  return SMMC_OK;
}

int smmc_engine_simulate(smmc_engine *e, const smmc_sim *sim, float *d_final, float *d_chunk_mean,
                         float *d_chunk_var, void *d_stats) {
  int rc = check_sim(e, sim);
  if (rc) return rc;
  if ((reinterpret_cast<uintptr_t>(d_final) | reinterpret_cast<uintptr_t>(d_chunk_mean) |
       reinterpret_cast<uintptr_t>(d_chunk_var)) & 3u)
    return fail(SMMC_ERR_INVALID, "d_final, d_chunk_mean and d_chunk_var must be 4-byte aligned");
  if (reinterpret_cast<uintptr_t>(d_stats) & 7u) return fail(SMMC_ERR_INVALID, "d_stats must be 8-byte aligned");
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  return enqueue_simulation(e, sim, d_final, d_chunk_mean, d_chunk_var, d_stats);
}

int smmc_engine_simulate_keepdata(smmc_engine *e, const smmc_sim *sim, float *d_traj, float *d_final) {
  int rc = check_sim(e, sim);
  if (rc) return rc;
  if (!d_traj) return fail(SMMC_ERR_INVALID, "d_traj is NULL");
  if ((reinterpret_cast<uintptr_t>(d_traj) | reinterpret_cast<uintptr_t>(d_final)) & 3u)
    return fail(SMMC_ERR_INVALID, "d_traj and d_final must be 4-byte aligned");
  if (sim->n_periods >= (1u << 24)) return fail(SMMC_ERR_INVALID, "keepdata supports n_periods < 2^24");
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  if (sim->n_paths == 0) return SMMC_OK;
  if (sim->flags & SMMC_FLAG_STREAM_REF)  // the reference's own stream: its kernels keep the trajectories themselves
    return enqueue_ref_simulation(e, sim, d_final, nullptr, nullptr, nullptr, d_traj);
  float unused_lo, unused_hi;
  const bool exact_div = divide_kind(e, sim, false, &unused_lo, &unused_hi) != SMMC_DIV_FAST;

  // Two kernels (smmc_kernels.hip).  The comb form takes whole 2048-row super-chunks when the row
  // length is odd by construction (n_periods a multiple of the draws per Philox block) and there is
  // enough of it to fill the chip; the tile form takes the rest (< 2048 rows), or everything.
  // SMMC_KEEPDATA_KERNEL=tile|comb forces one (results never depend on it).
  const uint32_t draws = smmc::keepdata_draws(sim->mode == SMMC_MODE_TABLE ? e->table_len : 0u);
  const uint64_t row_len = static_cast<uint64_t>(sim->n_periods) + 1;
  uint64_t n_super = sim->n_paths / 2048u;
  const bool comb_fits = sim->n_periods >= 64u && sim->n_periods % draws == 0u && row_len <= (1u << 20) && n_super >= 1;
  // waves per workgroup of the comb form: what fits the CU's LDS beside the tables, one workgroup per
  // CU; table mode (tiles of 40 columns: 14 would fit) runs 12, three per SIMD: 3 % faster than 14
  int comb_waves = 0;
  if (comb_fits) {
    const size_t lds_cu = 160u * 1024u;
    const uint32_t tl = sim->mode == SMMC_MODE_TABLE ? e->table_len : 0u;
    const int strm = (sim->flags & SMMC_FLAG_STREAM_V2) ? 2 : 3;
    const size_t fixed = smmc::keepdata_comb_lds_bytes(tl, 0, strm);
    const size_t per_wave = smmc::keepdata_comb_lds_bytes(tl, 1, strm) - fixed;
    comb_waves = fixed < lds_cu ? static_cast<int>(std::min<size_t>((lds_cu - fixed) / per_wave, 16)) : 0;
    if (sim->mode == SMMC_MODE_TABLE && comb_waves > 12) comb_waves = 12;
    if (const char *env = std::getenv("SMMC_KEEPDATA_COMB_WAVES")) {  // tuning knob
      const long v = std::strtol(env, nullptr, 10);
      if (v >= 1 && v <= comb_waves) comb_waves = static_cast<int>(v);
    }
  }
  // the comb form needs enough chunks (64 rows each) to keep its waves level: below ~4 per wave
  // (1e6 rows) the tile kernel's finer grain wins (0.20 vs 0.26 ms at 6e5 rows of 361 values; at
  // 1.5e6 rows of 1001 values the comb form is ahead, 1.41 vs 1.58 ms in Gaussian mode)
  bool comb = comb_fits && comb_waves >= 1 &&
              n_super * 32u >= 4ull * e->compute_units * static_cast<uint64_t>(comb_waves);
  if (const char *env = std::getenv("SMMC_KEEPDATA_KERNEL")) {
    if (!std::strcmp(env, "tile")) comb = false;
    if (!std::strcmp(env, "comb")) comb = comb_fits && comb_waves >= 1;
  }
  const uint64_t n_comb = comb ? n_super * 2048u : 0u;
  // Everything that can fail is decided BEFORE the first launch and before the timing pair opens
  // (a failing tile set-up used to leave the comb part written and a start event without its stop).
  // Tile form (the rows the comb form does not take): tuning knobs, results do not depend on them:
  // columns per LDS tile (16 | 32), waves per workgroup, workgroups per CU.  Defaults measured with
  // tools/kd_ab.py: 32 columns (16 writes half lines); table mode 4 waves, three workgroups per CU (8 or
  // 12 waves 1-5 % slower, 3 / 5 / 6 waves -- SIMDs unevenly filled -- 5-25 % slower); Gaussian mode,
  // where the Box-Muller tables take 18.9 KB of every workgroup's LDS, one 12-wave workgroup per CU
  // (2-15 % faster than two of 4); grid = what is resident (a wave strides over its 64-path chunks).
  int tile = 32, tile_waves = 0;
  uint32_t tile_grid = 0;
  if (n_comb < sim->n_paths) {
    const uint32_t tl = sim->mode == SMMC_MODE_TABLE ? e->table_len : 0u;
    const int strm = (sim->flags & SMMC_FLAG_STREAM_V2) ? 2 : 3;
    if (const char *env = std::getenv("SMMC_KEEPDATA_TILE")) {
      const long v = std::strtol(env, nullptr, 10);
      if (v == 16 || v == 32) tile = static_cast<int>(v);
    }
    const size_t lds_cu = 160u * 1024u;
    const size_t fixed = smmc::keepdata_lds_bytes(tl, tile, 0, strm);
    const size_t per_wave = smmc::keepdata_lds_bytes(tl, tile, 1, strm) - fixed;
    const int fit = fixed < lds_cu ? static_cast<int>(std::min<size_t>((lds_cu - fixed) / per_wave, 12)) : 0;
    if (fit < 1) return fail(SMMC_ERR_INVALID, "keepdata: the table leaves no LDS for a tile");
    tile_waves = sim->mode == SMMC_MODE_TABLE ? std::min(fit, 4) : (fit >= 12 ? 12 : fit >= 8 ? 8 : std::min(fit, 4));
    if (const char *env = std::getenv("SMMC_KEEPDATA_WAVES")) {
      const long v = std::strtol(env, nullptr, 10);
      if (v >= 1 && v <= fit) tile_waves = static_cast<int>(v);
    }
    const size_t lds = smmc::keepdata_lds_bytes(tl, tile, tile_waves, strm);
    const uint64_t n_wave_chunks = (sim->n_paths - n_comb + 63) / 64;
    const uint32_t resident = static_cast<uint32_t>(std::max<size_t>(lds_cu / lds, 1));
    const uint32_t per_cu = e->keepdata_blocks_per_cu ? e->keepdata_blocks_per_cu : resident;
    tile_grid = static_cast<uint32_t>(
        std::min<uint64_t>((n_wave_chunks + tile_waves - 1) / tile_waves, static_cast<uint64_t>(e->compute_units) * per_cu));
  }
  if (comb && !e->d_work_counter)
    SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_work_counter), sizeof(unsigned long long)));
  rc = timing_begin(e);
  if (rc) return rc;
  // from here on a failure closes the timing pair before it returns
  auto bail = [&](hipError_t err, const char *what) {
    (void)timing_end(e);
    return fail(SMMC_ERR_HIP, "%s failed: %s", what, hipGetErrorString(err));
  };
  if (comb) {
    smmc::KernelArgs a = make_args(e, sim);
    a.d_traj = d_traj;
    a.d_final = nullptr;
    a.n_paths = n_comb;
    const int waves = comb_waves;
    // rows per stream.  More rows cost fewer extra columns (4.3 % at K = 1, 2.1 % at K = 2 for P = 360)
    // but make the chunks a wave takes coarser.  Measured (tools/kd_ab.py, profiles/r02/keepdata_ab_stream_v3.txt):
    // Gaussian 4e6 x 361: K = 1 1.141 ms, K = 2 1.150, K = 4 1.159; 1.5e6 x 1001: 1.293 / 1.324 / 1.344;
    // table mode 1.095 / 1.103 and 1.212 / 1.236.  (With the costlier draws of stream v2 K = 2 won the
    // first shape by 1.3 %.)
    uint32_t k_rows = 1;
    if (const char *env = std::getenv("SMMC_KEEPDATA_K")) {  // tuning knob: 1, 2, 4, 8, 16 or 32
      const long v = std::strtol(env, nullptr, 10);
      if (v >= 1 && v <= 32 && (v & (v - 1)) == 0) k_rows = static_cast<uint32_t>(v);
    }
    const uint64_t n_wave_chunks = n_super * (32u / k_rows);
    const uint32_t per_cu = e->keepdata_blocks_per_cu ? e->keepdata_blocks_per_cu : 1u;
    const uint32_t cgrid = static_cast<uint32_t>(
        std::min<uint64_t>((n_wave_chunks + waves - 1) / waves, static_cast<uint64_t>(e->compute_units) * per_cu));
    // Philox blocks drawn together per step (instruction-level parallelism at < 4 waves per SIMD)
    int per_step = (sim->n_periods / draws) % 2u == 0u ? 2 : 1;
    if (const char *env = std::getenv("SMMC_KEEPDATA_COMB_ILP")) {  // tuning knob
      if (!std::strcmp(env, "1")) per_step = 1;
    }
    const hipError_t err = smmc::launch_keepdata_comb(a, exact_div, per_step, k_rows, n_wave_chunks, sim->n_paths, waves, cgrid,
                                                      e->d_work_counter, e->stream);
    if (err != hipSuccess) return bail(err, "launch_keepdata_comb");
  }
  if (n_comb < sim->n_paths) {
    smmc_sim rest = *sim;
    rest.first_path = sim->first_path + n_comb;
    rest.n_paths = sim->n_paths - n_comb;
    smmc::KernelArgs a = make_args(e, &rest);
    a.d_traj = d_traj + n_comb * row_len;
    a.d_final = (d_final && !comb) ? d_final : nullptr;
    const hipError_t err = smmc::launch_keepdata(a, exact_div, tile, tile_waves, tile_grid, e->stream);
    if (err != hipSuccess) return bail(err, "launch_keepdata");
  }
  if (comb && d_final) {  // the comb form leaves the final values to a gather of the last column
    const uint32_t fgrid = static_cast<uint32_t>(std::min<uint64_t>((sim->n_paths + smmc::kBlock - 1) / smmc::kBlock, e->max_grid));
    const hipError_t err = smmc::launch_final_column(d_traj, sim->n_paths, static_cast<uint32_t>(row_len), d_final, fgrid, e->stream);
    if (err != hipSuccess) return bail(err, "launch_final_column");
  }
  return timing_end(e);
}

int smmc_engine_sync(smmc_engine *e) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  DeviceGuard guard(e->device);
  SMMC_HIP(hipStreamSynchronize(e->stream));
  return SMMC_OK;
}

}  // extern "C"

namespace {

// The two device staging buffers of the host pipeline, at least `chunk` floats each.  Device must be current.
int reserve_staging(smmc_engine *e, uint64_t chunk) {
  if (e->stage_paths >= chunk) return SMMC_OK;
  for (int i = 0; i < 2; ++i) {
    if (e->d_stage[i]) SMMC_HIP(hipFree(e->d_stage[i]));
    e->d_stage[i] = nullptr;
  }
  e->stage_paths = 0;
  for (int i = 0; i < 2; ++i) SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_stage[i]), sizeof(float) * chunk));
  e->stage_paths = chunk;
  return SMMC_OK;
}

bool verbose_env() {
  const char *env = std::getenv("SMMC_VERBOSE");
  return env && *env && *env != '0';
}

// Is the byte at p in memory the runtime already knows as page-locked (hipHostMalloc'd or registered)?
bool is_pinned(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type != hipMemoryTypeUnregistered) return true;
  (void)hipGetLastError();
  return false;
}

}  // namespace

namespace smmc {
PinPolicy pin_policy_from_env() {
  const char *env = std::getenv("SMMC_PIN_HOST");
  if (!env || !std::strcmp(env, "1") || !std::strcmp(env, "whole")) return kPinWhole;
  if (!std::strcmp(env, "chunk")) return kPinChunk;
  return kPinNever;
}
bool host_range_is_pinned(const void *p, uint64_t bytes) {
  return p && bytes && is_pinned(p) && is_pinned(static_cast<const char *>(p) + (bytes - 1));
}
}  // namespace smmc

namespace {

// Page-locks whole pages [lo, hi) for the lifetime of the object, so that D2H copies into them run at the
// pinned rate without the runtime's staging.  Failure is not an error -- the copy then takes the pageable
// path -- but it is reported under SMMC_VERBOSE (a silently degraded pinned path).
struct HostPin {
  void *p = nullptr;
  hipStream_t drain = nullptr;  // stream whose copies may still target the range when an error unwinds
  bool pin_pages(uintptr_t lo, uintptr_t hi) {
    release();
    if (hi <= lo) return true;
    if (hipHostRegister(reinterpret_cast<void *>(lo), hi - lo, hipHostRegisterDefault) != hipSuccess) {
      if (verbose_env())
        std::fprintf(stderr, "smmc: hipHostRegister of %zu bytes at %p failed (%s): this part is copied through the pageable path\n",
                     static_cast<size_t>(hi - lo), reinterpret_cast<void *>(lo), hipGetErrorString(hipGetLastError()));
      else
        (void)hipGetLastError();
      return false;
    }
    p = reinterpret_cast<void *>(lo);
    return true;
  }
  void release() {
    if (p) (void)hipHostUnregister(p);
    p = nullptr;
  }
  ~HostPin() {
    if (p && drain) (void)hipStreamSynchronize(drain);
    release();
  }
};

}  // namespace

extern "C" {

int smmc_engine_prepare_host(smmc_engine *e, uint64_t n_paths) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  return reserve_staging(e, std::min<uint64_t>(std::max<uint64_t>(n_paths, 1), e->host_chunk_paths));
}

namespace {
std::mutex g_registered_mutex;
std::vector<std::pair<void *, void *>> g_registered;  // (caller's pointer, page-aligned registered base)
}  // namespace

int smmc_host_register(void *host_ptr, uint64_t bytes) {
  if (!host_ptr || bytes == 0) return fail(SMMC_ERR_INVALID, "empty host buffer");
  if (smmc::host_range_is_pinned(host_ptr, bytes)) return SMMC_OK;
  const uintptr_t page = 4096, lo = reinterpret_cast<uintptr_t>(host_ptr) & ~(page - 1),
                  hi = (reinterpret_cast<uintptr_t>(host_ptr) + bytes + page - 1) & ~(page - 1);
  SMMC_HIP(hipHostRegister(reinterpret_cast<void *>(lo), hi - lo, hipHostRegisterPortable));
  std::lock_guard<std::mutex> lock(g_registered_mutex);
  g_registered.emplace_back(host_ptr, reinterpret_cast<void *>(lo));
  return SMMC_OK;
}

int smmc_host_unregister(void *host_ptr) {
  void *base = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_registered_mutex);
    for (size_t i = 0; i < g_registered.size(); ++i)
      if (g_registered[i].first == host_ptr) {
        base = g_registered[i].second;
        g_registered.erase(g_registered.begin() + static_cast<long>(i));
        break;
      }
  }
  if (!base) return SMMC_OK;  // not registered by smmc_host_register (or pinned by the caller): nothing to undo
  SMMC_HIP(hipHostUnregister(base));
  return SMMC_OK;
}

int smmc_engine_simulate_to_host(smmc_engine *e, const smmc_sim *sim, float *host_final,
                                 float *host_chunk_mean, float *host_chunk_var, volatile int64_t *progress,
                                 smmc_stats *stats, uint64_t *hist) {
  int rc = check_sim(e, sim);
  if (rc) return rc;
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  const uint64_t n = sim->n_paths;
  PhaseLog phase("simulate_to_host");
  phase.on = phase.on && !(sim->flags & SMMC_FLAG_QUIET);
  const bool polled = progress != nullptr || e->progress_fn != nullptr;
  // chunk: a multiple of 1024 paths (4 KiB of floats: chunks of a page-aligned buffer do not share
  // pages).  64 MiB of floats by default; when a caller polls the progress counter (the reference
  // advances it every 1000 paths, src/simulations.cpp:254, and its GUIs redraw from it,
  // examples/visualize_returns_cpu_v2.cpp:360-376) about 16 steps per run, at least 2^16 paths -- the
  // 1e6 paths of BASELINE configs[0] advance in 16 steps (a 2^16-path launch still fills the chip:
  // 256 workgroups)
  uint64_t chunk_max = e->host_chunk_paths;
  if (polled && e->host_chunk_paths == kHostChunkPaths) {
    const uint64_t sixteenth = (n / 16 + 1023) / 1024 * 1024;
    chunk_max = std::min<uint64_t>(kHostChunkPaths, std::max<uint64_t>(kProgressChunkMin, sixteenth));
  }
  const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(n, 1), chunk_max);
  const uint64_t n_chunks = (n + chunk - 1) / chunk;
  const bool want_stats = stats != nullptr || hist != nullptr;
  const bool want_cs = host_chunk_mean != nullptr || host_chunk_var != nullptr;
  const size_t rec = smmc_stats_bytes(sim->n_bins);
  const uint64_t cs_per_chunk = (chunk + smmc::kBlock - 1) / smmc::kBlock;

  if (!e->copy_stream) SMMC_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    if (!e->ev_compute[i]) SMMC_HIP(hipEventCreateWithFlags(&e->ev_compute[i], hipEventDisableTiming));
    if (!e->ev_copy[i]) SMMC_HIP(hipEventCreateWithFlags(&e->ev_copy[i], hipEventDisableTiming));
  }
  if (host_final) {
    rc = reserve_staging(e, chunk);
    if (rc) return rc;
  }
  if (want_cs && e->stage_cs < cs_per_chunk) {
    for (int i = 0; i < 2; ++i) {
      if (e->d_stage_cs[i]) SMMC_HIP(hipFree(e->d_stage_cs[i]));
      e->d_stage_cs[i] = nullptr;
    }
    e->stage_cs = 0;
    for (int i = 0; i < 2; ++i)
      SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_stage_cs[i]), sizeof(float) * 2 * cs_per_chunk));
    e->stage_cs = cs_per_chunk;
  }
  if (want_stats && e->stage_stats_bytes < rec * std::max<uint64_t>(n_chunks, 1)) {
    if (e->d_stage_stats) SMMC_HIP(hipFree(e->d_stage_stats));
    e->d_stage_stats = nullptr;
    e->stage_stats_bytes = 0;
    SMMC_HIP(hipMalloc(&e->d_stage_stats, rec * std::max<uint64_t>(n_chunks, 1)));
    e->stage_stats_bytes = rec * std::max<uint64_t>(n_chunks, 1);
  }
  phase.mark("side stream, events, staging buffers");
  auto report = [&](uint64_t done) {
    if (progress) __atomic_store_n(const_cast<int64_t *>(progress), static_cast<int64_t>(done), __ATOMIC_RELEASE);
    if (e->progress_fn) e->progress_fn(e->progress_user, static_cast<int64_t>(done));
  };
  report(0);

  // Pinning the caller's result buffer (SMMC_PIN_HOST=whole|chunk|0; DESIGN.md section 6 has the
  // measurements).  "whole" (default): one hipHostRegister over all of host_final before the first chunk;
  // "chunk": chunk c + 1 is registered by this host thread while chunk c computes, and chunk c is
  // released once its copy has finished.  Buffers that are pinned already are left alone.
  // Registration is by whole pages with ONE owner per page (chunks are multiples of 4 KiB but
  // host_final need not be page aligned, so neighbouring chunks share a page; registering it twice fails
  // and that chunk silently took the pageable path): the page in which chunk c begins belongs to chunk c,
  // chunk c's registration ends where chunk c + 1's begins, the last one runs to the end of the buffer's
  // last page.  The copy of chunk c is enqueued while chunks c - 1, c and c + 1 are registered, so every
  // page it touches is pinned.  A copy may not span two registrations (hipMemcpyAsync: invalid argument),
  // so a chunk's copy is cut where the next owner's pages begin: the tail piece (< 4 KiB) is its own copy.
  HostPin pin_all, pin_chunk[3];
  pin_all.drain = pin_chunk[0].drain = pin_chunk[1].drain = pin_chunk[2].drain = e->copy_stream;
  const bool big_enough = host_final && sizeof(float) * n >= e->pin_min_bytes && !(sim->flags & SMMC_FLAG_HOST_NOPIN);
  const bool already = big_enough && smmc::host_range_is_pinned(host_final, sizeof(float) * n);
  const bool pin_whole = big_enough && !already && e->pin_policy == smmc::kPinWhole;
  const bool pin_chunks = big_enough && !already && e->pin_policy == smmc::kPinChunk && chunk >= 2048;  // chunks span whole pages
  const uintptr_t page = 4096;
  auto page_floor = [&](const float *q) { return reinterpret_cast<uintptr_t>(q) & ~(page - 1); };
  auto pin_chunk_c = [&](uint64_t c) {  // the pages chunk c owns
    const uintptr_t lo = page_floor(host_final + c * chunk);
    const uintptr_t hi = c + 1 < n_chunks ? page_floor(host_final + (c + 1) * chunk)
                                          : (reinterpret_cast<uintptr_t>(host_final + n) + page - 1) & ~(page - 1);
    (void)pin_chunk[c % 3].pin_pages(lo, hi);
  };
  if (pin_whole)
    (void)pin_all.pin_pages(page_floor(host_final), (reinterpret_cast<uintptr_t>(host_final + n) + page - 1) & ~(page - 1));
  if (pin_chunks) {
    pin_chunk_c(0);
    if (n_chunks > 1) pin_chunk_c(1);
  }

  phase.mark("result buffer registered");
  // Pipeline: the kernel of chunk c (engine stream) overlaps the D2H copies of chunk
  // c - 1 (copy stream).  Buffer b = c & 1 is reused once its copies have finished.
  const bool copies = host_final || want_cs;
  for (uint64_t c = 0; c < n_chunks; ++c) {
    const int b = static_cast<int>(c & 1);
    smmc_sim part = *sim;
    part.first_path = sim->first_path + c * chunk;
    part.n_paths = std::min<uint64_t>(chunk, n - c * chunk);
    const uint64_t cs_here = (part.n_paths + smmc::kBlock - 1) / smmc::kBlock;
    if (c >= 2 && copies) SMMC_HIP(hipStreamWaitEvent(e->stream, e->ev_copy[b], 0));
    // Host-side wait (progress reports, chunk registration): for the copy of chunk c - 2, BEFORE chunk c is
    // enqueued -- the event the stream itself waits on for its staging buffer.  The host then stays two
    // chunks ahead of the device and the next kernel is always queued when one finishes (waiting for chunk
    // c - 1 after enqueuing chunk c, as round 2 did, woke the host up when kernel c was nearly over:
    // 1e8 x 360 polled paths took 17 ms instead of 10).
    if ((polled || pin_chunks) && copies && c >= 2) {
      SMMC_HIP(hipEventSynchronize(e->ev_copy[b]));  // chunks 0 .. c - 2 are in the caller's memory
      if (polled) report((c - 1) * chunk);
      if (pin_chunks) pin_chunk[(c - 2) % 3].release();
    }
    if (pin_chunks && c >= 1 && c + 1 < n_chunks) pin_chunk_c(c + 1);  // chunks c - 1, c, c + 1 registered while copy c is enqueued
    void *d_rec = want_stats ? static_cast<char *>(e->d_stage_stats) + rec * c : nullptr;
    float *d_cm = want_cs ? e->d_stage_cs[b] : nullptr;
    float *d_cv = want_cs ? e->d_stage_cs[b] + cs_per_chunk : nullptr;
    rc = enqueue_simulation(e, &part, host_final ? e->d_stage[b] : nullptr, d_cm, d_cv, d_rec);
    if (rc) return rc;
    if (copies) {
      SMMC_HIP(hipEventRecord(e->ev_compute[b], e->stream));
      SMMC_HIP(hipStreamWaitEvent(e->copy_stream, e->ev_compute[b], 0));
      if (host_final) {
        char *dst = reinterpret_cast<char *>(host_final + c * chunk);
        const char *src = reinterpret_cast<const char *>(e->d_stage[b]);
        const size_t bytes = sizeof(float) * part.n_paths;
        size_t head = bytes;
        if (pin_chunks && c + 1 < n_chunks) {  // the last bytes lie in the page chunk c + 1 owns
          const uintptr_t cut = page_floor(host_final + (c + 1) * chunk);
          if (cut > reinterpret_cast<uintptr_t>(dst) && cut < reinterpret_cast<uintptr_t>(dst) + bytes)
            head = cut - reinterpret_cast<uintptr_t>(dst);
        }
        SMMC_HIP(hipMemcpyAsync(dst, src, head, hipMemcpyDeviceToHost, e->copy_stream));
        if (head < bytes) SMMC_HIP(hipMemcpyAsync(dst + head, src + head, bytes - head, hipMemcpyDeviceToHost, e->copy_stream));
      }
      if (host_chunk_mean)
        SMMC_HIP(hipMemcpyAsync(host_chunk_mean + c * cs_per_chunk, d_cm, sizeof(float) * cs_here,
                                hipMemcpyDeviceToHost, e->copy_stream));
      if (host_chunk_var)
        SMMC_HIP(hipMemcpyAsync(host_chunk_var + c * cs_per_chunk, d_cv, sizeof(float) * cs_here,
                                hipMemcpyDeviceToHost, e->copy_stream));
      SMMC_HIP(hipEventRecord(e->ev_copy[b], e->copy_stream));

    } else if (polled) {
      // nothing is copied per chunk: chunks 0 .. c - 2 are finished when the kernel of c - 2 is
      if (c >= 2) {
        SMMC_HIP(hipEventSynchronize(e->ev_compute[b]));
        report((c - 1) * chunk);
      }
      SMMC_HIP(hipEventRecord(e->ev_compute[b], e->stream));
    }
  }
  phase.mark("all chunks enqueued");
  SMMC_HIP(hipStreamSynchronize(e->stream));
  phase.mark("kernels done");
  if (copies) SMMC_HIP(hipStreamSynchronize(e->copy_stream));
  phase.mark("copies done");
  for (HostPin &hp : pin_chunk) hp.release();
  pin_all.release();
  phase.mark("registration released");

  if (want_stats) {
    std::vector<char> all(rec * std::max<uint64_t>(n_chunks, 1));
    std::vector<char> acc(rec, 0);
    smmc_stats *h = reinterpret_cast<smmc_stats *>(acc.data());
    h->min = std::numeric_limits<float>::infinity();
    h->max = -std::numeric_limits<float>::infinity();
    h->n_bins = sim->n_bins;
    if (n_chunks) {
      SMMC_HIP(hipMemcpy(all.data(), e->d_stage_stats, rec * n_chunks, hipMemcpyDeviceToHost));
      for (uint64_t c = 0; c < n_chunks; ++c) {
        rc = smmc_stats_merge(acc.data(), all.data() + rec * c);
        if (rc) return rc;
      }
    }
    if (stats) *stats = *h;
    if (hist && sim->n_bins) std::memcpy(hist, acc.data() + sizeof(smmc_stats), sizeof(uint64_t) * sim->n_bins);
  }
  report(n);
  return SMMC_OK;
}

int smmc_engine_simulate_keepdata_to_host(smmc_engine *e, const smmc_sim *sim, float *host_traj, float *host_final) {
  int rc = check_sim(e, sim);
  if (rc) return rc;
  if (!host_traj) return fail(SMMC_ERR_INVALID, "host_traj is NULL");
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  const uint64_t row = static_cast<uint64_t>(sim->n_periods) + 1;
  // trajectories of one slice: at most ~1 GiB on the device, a multiple of 256 paths
  uint64_t slice = std::max<uint64_t>((1ull << 28) / row, 1);
  slice = std::max<uint64_t>(slice / smmc::kBlock * smmc::kBlock, smmc::kBlock);
  slice = std::min<uint64_t>(slice, std::max<uint64_t>(sim->n_paths, 1));
  float *d_traj = nullptr, *d_fin = nullptr;
  SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&d_traj), sizeof(float) * slice * row));
  hipError_t err = hipMalloc(reinterpret_cast<void **>(&d_fin), sizeof(float) * slice);
  for (uint64_t first = 0; err == hipSuccess && first < sim->n_paths; first += slice) {
    smmc_sim part = *sim;
    part.first_path = sim->first_path + first;
    part.n_paths = std::min<uint64_t>(slice, sim->n_paths - first);
    rc = smmc_engine_simulate_keepdata(e, &part, d_traj, d_fin);
    if (rc) break;
    err = hipMemcpyAsync(host_traj + first * row, d_traj, sizeof(float) * part.n_paths * row, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess && host_final)
      err = hipMemcpyAsync(host_final + first, d_fin, sizeof(float) * part.n_paths, hipMemcpyDeviceToHost, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  }
  (void)hipFree(d_traj);
  if (d_fin) (void)hipFree(d_fin);
  if (rc) return rc;
  if (err != hipSuccess) return fail(SMMC_ERR_HIP, "keepdata_to_host failed: %s", hipGetErrorString(err));
  return SMMC_OK;
}

int smmc_engine_values_stats(smmc_engine *e, const float *d_values, uint64_t n, float below_threshold,
                             uint32_t n_bins, float hist_lo, float hist_hi, void *d_stats) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (!d_stats) return fail(SMMC_ERR_INVALID, "d_stats is NULL");
  if (n && !d_values) return fail(SMMC_ERR_INVALID, "d_values is NULL");
  if (n_bins > SMMC_MAX_BINS) return fail(SMMC_ERR_INVALID, "n_bins %u exceeds SMMC_MAX_BINS %d", n_bins, SMMC_MAX_BINS);
  if (n_bins && (!(hist_lo < hist_hi) || !std::isfinite(hist_lo) || !std::isfinite(hist_hi)))
    return fail(SMMC_ERR_INVALID, "histogram range must be finite with lo < hi");
  if (reinterpret_cast<uintptr_t>(d_values) & 3u) return fail(SMMC_ERR_INVALID, "d_values must be 4-byte aligned");
  if (reinterpret_cast<uintptr_t>(d_stats) & 7u) return fail(SMMC_ERR_INVALID, "d_stats must be 8-byte aligned");
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  return values_stats_enqueue(e, d_values, n, below_threshold, n_bins, hist_lo, hist_hi, d_stats, true);
}

int smmc_engine_order_statistics(smmc_engine *e, const float *d_values, uint64_t n, const uint64_t *ranks,
                                 uint32_t n_ranks, float *host_out) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (!d_values || !ranks || !host_out) return fail(SMMC_ERR_INVALID, "NULL argument");
  if (n_ranks == 0 || n_ranks > SMMC_MAX_RANKS) return fail(SMMC_ERR_INVALID, "n_ranks must be in [1, %d]", SMMC_MAX_RANKS);
  if (reinterpret_cast<uintptr_t>(d_values) & 3u) return fail(SMMC_ERR_INVALID, "d_values must be 4-byte aligned");
  for (uint32_t q = 0; q < n_ranks; ++q)
    if (ranks[q] >= n) return fail(SMMC_ERR_INVALID, "rank %llu is not below n = %llu", (unsigned long long)ranks[q], (unsigned long long)n);
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  const size_t hist_bytes = sizeof(unsigned long long) * smmc::kMaxRanks * 2048;
  if (!e->d_select) SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_select), sizeof(smmc::SelectState)));
  if (!e->d_radix_hist) {
    SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_radix_hist), hist_bytes));
    e->radix_dirty = true;
  }
  if (!e->d_select_out) SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_select_out), sizeof(float) * smmc::kMaxRanks));
  if (!e->h_select) SMMC_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->h_select), sizeof(smmc::SelectState), hipHostMallocDefault));
  if (e->radix_dirty) {
    SMMC_HIP(hipMemsetAsync(e->d_radix_hist, 0, hist_bytes, e->stream));
    e->radix_dirty = false;
  }
  // the state goes up from the engine's page-locked copy: every call ends with a synchronisation of the stream, so the
  // copy of the call before has long been read
  std::memset(e->h_select, 0, sizeof(smmc::SelectState));
  for (uint32_t q = 0; q < n_ranks; ++q) e->h_select->rank[q] = ranks[q];
  SMMC_HIP(hipMemcpyAsync(e->d_select, e->h_select, sizeof(smmc::SelectState), hipMemcpyHostToDevice, e->stream));
  // 1024-thread workgroups, four per CU of which two are resident (the LDS histograms allow no more);
  // the queued ones even out the CUs: 1e9 values 0.730 -> 0.699 ms per pass (2 -> 4 per CU)
  const uint64_t want = (n / 8 + 1023) / 1024;
  uint32_t radix_per_cu = 4;
  if (const char *env = std::getenv("SMMC_RADIX_BLOCKS_PER_CU")) {  // tuning knob
    const long v = std::strtol(env, nullptr, 10);
    if (v >= 1 && v <= 16) radix_per_cu = static_cast<uint32_t>(v);
  }
  const uint32_t grid = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint64_t>(want, 1), e->compute_units * radix_per_cu));
  e->radix_dirty = true;  // until the last pick of the call is in the queue: an error return in between leaves counts behind
  for (int pass = 0; pass < 3; ++pass) {
    int rc = timing_begin(e);
    if (rc) return rc;
    SMMC_HIP_TIMED(e, smmc::launch_radix_hist(d_values, n, pass, n_ranks, e->d_select, e->d_radix_hist, grid, e->stream));
    rc = timing_end(e);
    if (rc) return rc;
    SMMC_HIP(smmc::launch_radix_pick(pass, n_ranks, e->d_select, e->d_radix_hist, e->d_select_out, e->stream));
  }
  e->radix_dirty = false;
  SMMC_HIP(hipMemcpyAsync(host_out, e->d_select_out, sizeof(float) * n_ranks, hipMemcpyDeviceToHost, e->stream));
  SMMC_HIP(hipStreamSynchronize(e->stream));
  return SMMC_OK;
}

int smmc_engine_quartiles(smmc_engine *e, const float *d_values, uint64_t n, float host_out[5]) {
  if (n == 0) return fail(SMMC_ERR_INVALID, "quartiles of an empty array");
  // examples/visualize_returns_cpu_v2.cpp:96-98
  const uint64_t q1 = n / 4, q2 = n / 2, q3 = q1 + q2;
  const uint64_t ranks[5] = {0, q1, q2, q3 < n ? q3 : n - 1, n - 1};
  return smmc_engine_order_statistics(e, d_values, n, ranks, 5, host_out);
}

int smmc_engine_reduce_mean_host(smmc_engine *e, const float *host_values, uint64_t n, float *mean, double *sum) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (!mean) return fail(SMMC_ERR_INVALID, "mean is NULL");
  if (n == 0) return fail(SMMC_ERR_INVALID, "mean of an empty array");
  if (!host_values) return fail(SMMC_ERR_INVALID, "host_values is NULL");
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  const uint64_t chunk = std::min<uint64_t>(n, kReduceChunkValues);
  {
    const int rc = reserve_staging(e, chunk);
    if (rc) return rc;
  }
  if (!e->copy_stream) SMMC_HIP(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    if (!e->ev_compute[i]) SMMC_HIP(hipEventCreateWithFlags(&e->ev_compute[i], hipEventDisableTiming));
    if (!e->ev_copy[i]) SMMC_HIP(hipEventCreateWithFlags(&e->ev_copy[i], hipEventDisableTiming));
  }
  const uint64_t n_chunks = (n + chunk - 1) / chunk;
  const size_t rec = smmc_stats_bytes(0);
  if (e->stage_stats_bytes < rec * n_chunks) {
    if (e->d_stage_stats) SMMC_HIP(hipFree(e->d_stage_stats));
    e->d_stage_stats = nullptr;
    e->stage_stats_bytes = 0;
    SMMC_HIP(hipMalloc(&e->d_stage_stats, rec * n_chunks));
    e->stage_stats_bytes = rec * n_chunks;
  }
  // H2D of chunk c (copy stream) overlaps the reduction of chunk c - 1 (engine stream)
  for (uint64_t c = 0; c < n_chunks; ++c) {
    const int b = static_cast<int>(c & 1);
    const uint64_t count = std::min<uint64_t>(chunk, n - c * chunk);
    if (c >= 2) SMMC_HIP(hipStreamWaitEvent(e->copy_stream, e->ev_compute[b], 0));
    SMMC_HIP(hipMemcpyAsync(e->d_stage[b], host_values + c * chunk, sizeof(float) * count, hipMemcpyHostToDevice, e->copy_stream));
    SMMC_HIP(hipEventRecord(e->ev_copy[b], e->copy_stream));
    SMMC_HIP(hipStreamWaitEvent(e->stream, e->ev_copy[b], 0));
    int rc = smmc_engine_values_stats(e, e->d_stage[b], count, 0.f, 0, 0.f, 1.f, static_cast<char *>(e->d_stage_stats) + rec * c);
    if (rc) return rc;
    SMMC_HIP(hipEventRecord(e->ev_compute[b], e->stream));
  }
  std::vector<smmc_stats> recs(n_chunks);
  SMMC_HIP(hipMemcpyAsync(recs.data(), e->d_stage_stats, rec * n_chunks, hipMemcpyDeviceToHost, e->stream));
  SMMC_HIP(hipStreamSynchronize(e->stream));
  double total = 0.0;
  for (uint64_t c = 0; c < n_chunks; ++c) total += recs[c].sum;  // chunk order
  if (sum) *sum = total;
  *mean = static_cast<float>(total) / static_cast<float>(n);
  return SMMC_OK;
}

int smmc_engine_host_values_summary(smmc_engine *e, const float *host_values, uint64_t n, float below_threshold,
                                    uint32_t n_bins, float hist_lo, float hist_hi, smmc_stats *stats,
                                    uint64_t *hist, float quartiles[5]) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (n == 0 || !host_values) return fail(SMMC_ERR_INVALID, "empty input");
  if (n_bins > SMMC_MAX_BINS) return fail(SMMC_ERR_INVALID, "n_bins %u exceeds SMMC_MAX_BINS %d", n_bins, SMMC_MAX_BINS);
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  float *d = nullptr;
  SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&d), sizeof(float) * n));
  int rc = SMMC_OK;
  hipError_t err = hipMemcpyAsync(d, host_values, sizeof(float) * n, hipMemcpyHostToDevice, e->stream);
  if (err == hipSuccess && !e->d_scratch_stats) err = hipMalloc(&e->d_scratch_stats, smmc_stats_bytes(SMMC_MAX_BINS));
  if (err == hipSuccess && (stats || hist)) {
    rc = smmc_engine_values_stats(e, d, n, below_threshold, n_bins, hist_lo, hist_hi, e->d_scratch_stats);
    if (rc == SMMC_OK) {
      std::vector<char> rec(smmc_stats_bytes(n_bins));
      err = hipMemcpyAsync(rec.data(), e->d_scratch_stats, rec.size(), hipMemcpyDeviceToHost, e->stream);
      if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
      if (err == hipSuccess) {
        if (stats) std::memcpy(stats, rec.data(), sizeof(smmc_stats));
        if (hist && n_bins) std::memcpy(hist, rec.data() + sizeof(smmc_stats), sizeof(uint64_t) * n_bins);
      }
    }
  }
  if (err == hipSuccess && rc == SMMC_OK && quartiles) rc = smmc_engine_quartiles(e, d, n, quartiles);
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  (void)hipFree(d);
  if (rc) return rc;
  if (err != hipSuccess) return fail(SMMC_ERR_HIP, "host_values_summary failed: %s", hipGetErrorString(err));
  return SMMC_OK;
}

int smmc_engine_set_stream(smmc_engine *e, void *stream) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (stream == SMMC_STREAM_NEW) return fail(SMMC_ERR_INVALID, "smmc_engine_set_stream takes a stream handle");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (s == e->stream) return SMMC_OK;
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  // work already enqueued keeps its order: the new stream waits for the old one's tail (the
  // engine's workspace -- partials, staging -- is shared between consecutive launches)
  if (!e->ev_order) SMMC_HIP(hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming));
  hipError_t err = hipEventRecord(e->ev_order, e->stream);
  if (err == hipSuccess) {
    SMMC_HIP(hipStreamWaitEvent(s, e->ev_order, 0));
  } else if (!e->own_stream) {
    // The stream the engine was bound to is not the engine's: the caller may have destroyed it since
    // (a bound stream has to outlive the NEXT set_stream only if work is still pending on it).  The
    // engine must not be stuck on it: drop the error, let the device drain -- whatever was enqueued
    // there has then finished, which is the ordering the event would have given -- and move on.
    (void)hipGetLastError();
    SMMC_HIP(hipDeviceSynchronize());
  } else {
    return fail(SMMC_ERR_HIP, "hipEventRecord on the engine's own stream failed: %s", hipGetErrorString(err));
  }
  if (e->own_stream) {
    SMMC_HIP(hipStreamSynchronize(e->stream));
    SMMC_HIP(hipStreamDestroy(e->stream));
    e->own_stream = false;
  }
  e->stream = s;
  return SMMC_OK;
}

int smmc_engine_get_stream(smmc_engine *e, void **stream) {
  if (!e || !stream) return fail(SMMC_ERR_INVALID, "NULL argument");
  *stream = static_cast<void *>(e->stream);
  return SMMC_OK;
}

int smmc_engine_wait_stream(smmc_engine *e, void *stream) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  hipStream_t other = static_cast<hipStream_t>(stream);
  if (other == e->stream) return SMMC_OK;
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  if (!e->ev_order) SMMC_HIP(hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming));
  SMMC_HIP(hipEventRecord(e->ev_order, other));
  SMMC_HIP(hipStreamWaitEvent(e->stream, e->ev_order, 0));
  return SMMC_OK;
}

int smmc_engine_release_to_stream(smmc_engine *e, void *stream) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  hipStream_t other = static_cast<hipStream_t>(stream);
  if (other == e->stream) return SMMC_OK;
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  if (!e->ev_order) SMMC_HIP(hipEventCreateWithFlags(&e->ev_order, hipEventDisableTiming));
  SMMC_HIP(hipEventRecord(e->ev_order, e->stream));
  SMMC_HIP(hipStreamWaitEvent(other, e->ev_order, 0));
  return SMMC_OK;
}

int smmc_engine_set_progress(smmc_engine *e, smmc_progress_fn fn, void *user) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  e->progress_fn = fn;
  e->progress_user = fn ? user : nullptr;
  return SMMC_OK;
}

int smmc_engine_timing(smmc_engine *e, int enable) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (enable && !e->d_clock) {
    DeviceGuard guard(e->device);
    if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
    SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&e->d_clock), 2 * sizeof(unsigned long long)));
    SMMC_HIP(hipMemsetAsync(e->d_clock, 0, 2 * sizeof(unsigned long long), e->stream));
  }
  e->timing = enable != 0;
  return SMMC_OK;
}

int smmc_engine_kernel_clock(smmc_engine *e, double *ghz) {
  if (!e || !ghz) return fail(SMMC_ERR_INVALID, "NULL argument");
  *ghz = 0.0;
  if (!e->d_clock) return SMMC_OK;  // timing was never enabled
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  unsigned long long h[2] = {0, 0};
  SMMC_HIP(hipMemcpyAsync(h, e->d_clock, sizeof h, hipMemcpyDeviceToHost, e->stream));
  SMMC_HIP(hipMemsetAsync(e->d_clock, 0, sizeof h, e->stream));
  SMMC_HIP(hipStreamSynchronize(e->stream));
  if (h[1]) *ghz = static_cast<double>(h[0]) / static_cast<double>(h[1]) * 0.1;  // s_memrealtime ticks at 100 MHz
  return SMMC_OK;
}

int smmc_engine_kernel_ms(smmc_engine *e, double *total_ms, uint32_t *launches) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  DeviceGuard guard(e->device);
  SMMC_HIP(hipStreamSynchronize(e->stream));
  double total = 0.0;
  for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
    float ms = 0.f;
    SMMC_HIP(hipEventElapsedTime(&ms, e->ev_pool[i], e->ev_pool[i + 1]));
    total += ms;
  }
  if (total_ms) *total_ms = total;
  if (launches) *launches = static_cast<uint32_t>(e->ev_used / 2);
  e->ev_used = 0;
  return SMMC_OK;
}

int smmc_engine_selftest(smmc_engine *e, uint32_t bits_lo, uint32_t bits_hi, uint64_t *div_mismatches) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (bits_hi < bits_lo) return fail(SMMC_ERR_INVALID, "empty bit-pattern range");
  DeviceGuard guard(e->device);
  if (!guard.ok) return fail(SMMC_ERR_HIP, "hipSetDevice(%d) failed", e->device);
  unsigned long long *d = nullptr, h = 0;
  SMMC_HIP(hipMalloc(reinterpret_cast<void **>(&d), sizeof h));
  hipError_t err = hipMemsetAsync(d, 0, sizeof h, e->stream);
  if (err == hipSuccess && bits_hi > bits_lo) err = smmc::launch_selftest(bits_lo, bits_hi, d, e->max_grid, e->stream);
  if (err == hipSuccess) err = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, e->stream);
  if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
  (void)hipFree(d);
  if (err != hipSuccess) return fail(SMMC_ERR_HIP, "self-test failed: %s", hipGetErrorString(err));
  if (div_mismatches) *div_mismatches = h;
  return SMMC_OK;
}

int smmc_engine_geometry(smmc_engine *e, uint32_t *grid, uint32_t *block, uint32_t *compute_units) {
  if (!e) return fail(SMMC_ERR_INVALID, "engine is NULL");
  if (grid) *grid = e->max_grid;
  if (block) *block = smmc::kBlock;
  if (compute_units) *compute_units = e->compute_units;
  return SMMC_OK;
}

int smmc_engine_divide_kind(smmc_engine *e, const smmc_sim *sim, int keepdata) {
  const int rc = check_sim(e, sim);
  if (rc) return rc;
  float lo, hi;
  return divide_kind(e, sim, keepdata == 0, &lo, &hi);
}

uint64_t smmc_stats_bytes(uint32_t n_bins) { return sizeof(smmc_stats) + sizeof(uint64_t) * n_bins; }

int smmc_stats_merge(void *dst_packed, const void *src_packed) {
  if (!dst_packed || !src_packed) return fail(SMMC_ERR_INVALID, "NULL statistics record");
  smmc_stats *d = static_cast<smmc_stats *>(dst_packed);
  const smmc_stats *s = static_cast<const smmc_stats *>(src_packed);
  if (d->n_bins != s->n_bins) return fail(SMMC_ERR_INVALID, "records have %u and %u bins", d->n_bins, s->n_bins);
  d->count += s->count;
  d->below += s->below;
  d->underflow += s->underflow;
  d->overflow += s->overflow;
  d->sum += s->sum;
  d->sumsq += s->sumsq;
  d->min = std::min(d->min, s->min);
  d->max = std::max(d->max, s->max);
  uint64_t *dh = reinterpret_cast<uint64_t *>(d + 1);
  const uint64_t *sh = reinterpret_cast<const uint64_t *>(s + 1);
  for (uint32_t i = 0; i < d->n_bins; ++i) dh[i] += sh[i];
  return SMMC_OK;
}

}  // extern "C"
