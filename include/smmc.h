/*
 * smmc.h -- C ABI of the MI355X Monte-Carlo returns engine (libsmmc_hip.so).
 *
 * This is the drop-in boundary for ONE path of matthijsvk/stock_market_monte_carlo:
 * the Monte-Carlo returns engine of src/simulations.cpp / src/simulations.cu.
 * Plain pointers and sizes only; no C++ or torch types.  Each entry point names
 * the reference interface it stands in for (paths relative to the reference
 * tree).  The C++ header include/stock_market_monte_carlo/simulations.h re-exports
 * the reference's own free-function signatures on top of these.
 *
 * Threading: an engine is bound to one device and one stream and is not
 * re-entrant; use one engine per host thread (engines are cheap).  Different
 * engines may be used concurrently.  No call exits the process: every failure
 * is a negative return code plus smmc_last_error() (thread-local text).
 *
 * Random stream ("counter stream v3"; round 1's stream v2 stays selectable
 * with SMMC_FLAG_STREAM_V2): Philox4x32-10, key =
 * the 64-bit seed, counter = (block of periods, global path id, mode).  A path's value
 * depends only on (seed, global path id, parameters), never on the launch
 * geometry, the shard it falls in or the number of GPUs.
 */
#ifndef SMMC_H
#define SMMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMMC_ABI_VERSION 4

/* return codes */
#define SMMC_OK 0
#define SMMC_ERR_INVALID (-1)   /* bad argument */
#define SMMC_ERR_HIP (-2)       /* a HIP runtime call failed */
#define SMMC_ERR_NO_DEVICE (-3) /* no usable gfx950 device */
#define SMMC_ERR_NOMEM (-4)

/* how a period's return is drawn */
#define SMMC_MODE_TABLE 0    /* i.i.d. with replacement from the returns table   */
#define SMMC_MODE_GAUSSIAN 1 /* N(gauss_mean, gauss_std), Box-Muller             */

/* smmc_sim.flags */
#define SMMC_FLAG_EXACT_DIV 1u /* force the IEEE divide kernel variant */
#define SMMC_FLAG_STREAM_V2 2u /* counter stream v2 (round 1's) instead of v3: its counter layout (both modes) and its Gaussian draw */
/* The reference CPU engine's OWN stream (src/simulations.cpp:240-252), table mode only: path id draws from
 * std::mt19937 seeded with (uint32_t)(seed + id) -- the reference seeds each path from a fresh
 * std::random_device and has no seed argument -- through libstdc++'s uniform_int_distribution<int>
 * (Lemire's map with rejection), then update_fund.  With the seeds the reference's generators got, the
 * final values (and, with the keepdata entries, the trajectories: mc_simulations_keepdata draws the same
 * way, src/simulations.cpp:175-186) are the reference's, bit for bit.  Statistics and chunk outputs are
 * formed from the final values by a second pass.  About 3x the arithmetic of the default stream at 360
 * periods and 4x at 1000 (the generator's 624 words of state per path are regenerated from seed chains, never
 * stored, for paths of up to 1816 periods; longer paths keep them in device memory). */
#define SMMC_FLAG_STREAM_REF 4u
#define SMMC_FLAG_QUIET 8u /* no SMMC_VERBOSE phase lines for this request (the drop-in's warm-up run) */
/* smmc_engine_simulate_to_host: leave host_final as it is -- no page-locking for the call whatever SMMC_PIN_HOST
 * says (the caller has pinned it, or has tried and failed: smmc_group_simulate registers the whole result once
 * for all its devices and passes this to every shard, so that neighbouring shards never register the page
 * their boundary falls in twice). */
#define SMMC_FLAG_HOST_NOPIN 16u

/* paths per chunk of the per-chunk mean/variance outputs: the reference's
 * THREADS_PER_BLOCK (src/simulations.cu:17), one (mean, variance) pair per block
 * in mc_simulations_gpu_kernel_reduceBlock (src/simulations.cu:240-246). */
#define SMMC_CHUNK 256

/* largest returns table an engine accepts (it is staged whole in LDS) */
#define SMMC_MAX_TABLE 16384
/* largest histogram */
#define SMMC_MAX_BINS 4096

typedef struct smmc_engine smmc_engine;

/* One simulation request: n_paths independent paths with global ids
 * first_path .. first_path + n_paths - 1, n_periods compounding steps each.
 * Mirrors the argument lists of mc_simulations (src/simulations.cpp:204-209) and
 * mc_simulations_gpu (src/simulations.cu:661-667) plus what the reference leaves
 * implicit (seed, draw mode) or computes on the host afterwards (statistics:
 * examples/benchmark_mc_gpu.cpp:7-41). */
typedef struct smmc_sim {
  uint32_t struct_size;   /* = sizeof(smmc_sim) */
  int32_t mode;           /* SMMC_MODE_*                                        */
  uint64_t seed;          /* Philox key                                         */
  uint64_t first_path;    /* global id of the first path (sharding offset)      */
  uint64_t n_paths;       /* max_n_simulations                                  */
  uint32_t n_periods;     /* n_periods                                          */
  float initial_capital;  /* initial_capital                                    */
  float gauss_mean;       /* percent per period (examples/monte_carlo_simulated.cpp:11) */
  float gauss_std;        /* percent per period (examples/monte_carlo_simulated.cpp:12) */
  uint32_t n_bins;        /* histogram buckets, 0 = none, <= SMMC_MAX_BINS      */
  float hist_lo, hist_hi; /* bucket range [lo, hi)                              */
  float below_threshold;  /* count of final values < this                       */
  uint32_t flags;         /* SMMC_FLAG_*                                        */
} smmc_sim;

/* Packed statistics record as the device writes it: this header followed by
 * n_bins uint64 bucket counts.  smmc_stats_bytes(n_bins) is its size.  All
 * integer fields are exact; sum/sumsq are double-precision sums of the float
 * final values in a fixed (launch-geometry dependent) order. */
typedef struct smmc_stats {
  uint64_t count;     /* paths simulated                                       */
  uint64_t below;     /* final value < below_threshold (benchmark_mc_gpu.cpp:30-41) */
  uint64_t underflow; /* final value < hist_lo                                 */
  uint64_t overflow;  /* final value >= hist_hi, or NaN                        */
  double sum;         /* sum of final values (benchmark_mc_gpu.cpp:13-17)      */
  double sumsq;       /* sum of squares                                        */
  float min, max;     /* +inf / -inf when count == 0                           */
  uint32_t n_bins;
  uint32_t reserved;
} smmc_stats;

/* ---- host scalar functions ------------------------------------------------ */

/* update_fund, src/simulations.cpp:14-16: fund * (100.0f + r) / 100 in binary32. */
float smmc_update_fund(float fund_value, float period_return);

/* __many_updates, src/simulations.cpp:18-22: totals[0] is read, totals[1..n] written. */
void smmc_many_updates(const float *returns, float *totals, uint32_t n_periods);

/* ---- library / device ------------------------------------------------------ */

int smmc_abi_version(void);
const char *smmc_last_error(void);
/* 64 hex digits: sha256 over the compiler flags and every source and header this library was built from
 * (stock_market_monte_carlo_amd/build.py: source_digest()).  The Python loader refuses a library whose digest
 * differs from the sources beside it; bench.py prints it; the reference has no counterpart (its build is
 * CMake's, CMakeLists.txt:99-103). */
const char *smmc_build_digest(void);

/* Number of visible HIP devices (0 and SMMC_OK when there is none). */
int smmc_device_count(int *count);

/* ---- engine ---------------------------------------------------------------- */

/* Pass as `stream` to make the engine create (and own) a non-blocking stream. */
#define SMMC_STREAM_NEW ((void *)(intptr_t)-1)

/* Binds an engine to `device`.  `stream` is the hipStream_t to launch on (e.g. the
 * caller's current torch stream); NULL is the device's default stream, as in every
 * HIP call; SMMC_STREAM_NEW asks for an engine-owned non-blocking stream.
 * Replaces the per-call cudaSetDevice/cudaMalloc/cudaFree plan of
 * create_plan_v2 (src/simulations.cu:568-574, 599-607, 632-637). */
int smmc_engine_create(int device, void *stream, smmc_engine **out);
void smmc_engine_destroy(smmc_engine *e);

/* Re-binds the engine to another stream of its device (e.g. the caller's CURRENT torch stream,
 * passed before every call).  Work already enqueued stays ordered before anything enqueued
 * afterwards.  An engine-owned stream is drained and destroyed.  A caller's stream must outlive the
 * work the engine enqueued on it; if it was destroyed after that work finished, the next set_stream
 * still succeeds (the event record on the dead handle fails, the engine lets the device drain and
 * adopts the new stream) -- every other call uses the stream bound last, so re-bind first.  smmc_engine_get_stream returns the
 * handle launches go to (so a caller can order its own streams against an engine-owned one). */
int smmc_engine_set_stream(smmc_engine *e, void *stream);
int smmc_engine_get_stream(smmc_engine *e, void **stream);
/* Ordering against another stream of the same device, for engines that keep their own stream:
 * wait_stream -- what the engine enqueues from now on runs after everything enqueued on `stream`
 * so far (call it before launching into buffers that were allocated or written on `stream`);
 * release_to_stream -- what is enqueued on `stream` from now on runs after everything the engine has
 * enqueued so far (call it before `stream` reads, frees or reuses the outputs).  Event-based, no
 * host synchronisation. */
int smmc_engine_wait_stream(smmc_engine *e, void *stream);
int smmc_engine_release_to_stream(smmc_engine *e, void *stream);

/* Uploads the historical-returns table (percent units, host memory).  Replaces
 * the H2D table copies at src/simulations.cu:382,451,525,617.  Waits for work already
 * enqueued on the engine stream; the host array may be reused on return. */
int smmc_engine_set_table(smmc_engine *e, const float *returns_percent, uint32_t n);

/* Enqueues one simulation on the engine stream and returns without waiting.
 * All output pointers are DEVICE pointers on the engine's device; any may be NULL:
 *   d_final       n_paths floats, final value of each path, coalesced
 *                 (totals of mc_simulations_gpu_kernel, src/simulations.cu:151)
 *   d_chunk_mean  ceil(n_paths / SMMC_CHUNK) floats, mean of each 256-path chunk
 *   d_chunk_var   same length, population variance of each chunk
 *                 (means/variances of the reduceBlock kernel, src/simulations.cu:240-246)
 *   d_stats       smmc_stats_bytes(sim->n_bins) bytes, packed statistics record, 8-byte aligned
 * Replaces mc_simulations_gpu_launcher / _reduceBlock_launcher
 * (src/simulations.cu:345-473). */
int smmc_engine_simulate(smmc_engine *e, const smmc_sim *sim, float *d_final, float *d_chunk_mean,
                         float *d_chunk_var, void *d_stats);

/* Same, but keeps every trajectory: d_traj is n_paths x (n_periods + 1) floats,
 * path-major (row i = the `values` vector of path i, values[0] = initial capital)
 * -- mc_data of mc_simulations_keepdata (src/simulations.cpp:139-186).  d_final
 * may be NULL.  Any 4-byte aligned d_traj; n_periods < 2^24 (SMMC_ERR_INVALID
 * otherwise). */
int smmc_engine_simulate_keepdata(smmc_engine *e, const smmc_sim *sim, float *d_traj, float *d_final);

/* Blocks until everything enqueued on the engine stream has finished. */
int smmc_engine_sync(smmc_engine *e);

/* Progress callback of the synchronous *_to_host entry points: called on the calling thread
 * with the number of paths whose results are in the caller's memory -- 0 at the start, after every
 * finished chunk, n_paths at the end.  The C++ drop-in layer stores it into the caller's
 * std::atomic<long> n_simulations (src/simulations.cpp:254; polled by
 * examples/visualize_returns_cpu_v2.cpp:360-376).  NULL clears it. */
typedef void (*smmc_progress_fn)(void *user, int64_t finished_paths);
int smmc_engine_set_progress(smmc_engine *e, smmc_progress_fn fn, void *user);

/* Simulates into HOST memory: outputs are produced in chunks of 2^22 paths and
 * copied back on a side stream while the next chunk computes (the async
 * cudaMemcpy pattern of mc_simulations_multi_gpu_launcher_async,
 * src/simulations.cu:615-626, without its extra host copy :643-644).
 * Host pointers (pinned or pageable), any may be NULL:
 *   host_final       n_paths floats
 *   host_chunk_mean  ceil(n_paths / SMMC_CHUNK) floats  (means of the reduceBlock API)
 *   host_chunk_var   same length                         (variances)
 *   progress         set (atomic release store) to the number of finished paths after every
 *                    chunk (the n_simulations counter of src/simulations.cpp:254); when it or a
 *                    progress callback is given, chunks shrink to about n_paths / 16 (at least
 *                    2^16 paths) so that a poller sees the run advance
 * Environment: SMMC_PIN_HOST=whole|chunk|0: a host_final of 32 MiB or more that is not pinned
 * already is page-locked (hipHostRegister, whole pages) for the duration of the call -- the whole
 * buffer up front (default: registration runs at 25-75 GB/s and lets the copies overlap the kernels),
 * or chunk by chunk one chunk ahead of the copies (every page has one owning chunk), or not at all; a
 * failed registration falls back to the pageable copy (reported under SMMC_VERBOSE).
 * SMMC_HOST_CHUNK_PATHS overrides the chunk length.  The per-path values, the chunk means / variances,
 * the counters, min / max and the histogram never depend on the chunk length (and so not on whether
 * progress is polled); sum and sumsq are double sums of the per-chunk records in chunk order and can
 * differ in their last bits between two chunkings.
 *   stats, hist      merged statistics header and n_bins bucket counts
 * Synchronous. */
int smmc_engine_simulate_to_host(smmc_engine *e, const smmc_sim *sim, float *host_final,
                                 float *host_chunk_mean, float *host_chunk_var, volatile int64_t *progress,
                                 smmc_stats *stats, uint64_t *hist);

/* Optional: allocates now what the next smmc_engine_simulate_to_host of up to n_paths final values will
 * need on the device (its two staging buffers), so that the first call of a process does not pay for it
 * -- e.g. while another host thread sizes the result buffer.  Results never depend on it. */
int smmc_engine_prepare_host(smmc_engine *e, uint64_t n_paths);

/* Page-locks / releases a caller's host buffer (hipHostRegister over its whole pages, visible to every
 * device): a buffer registered here is used as it is by smmc_engine_simulate_to_host and
 * smmc_group_simulate (no registration per call; 12-15 ms per 400 MB the first time).  A buffer that is
 * pinned already is left alone (SMMC_OK; smmc_host_unregister of it is then a no-op). */
int smmc_host_register(void *host_ptr, uint64_t bytes);
int smmc_host_unregister(void *host_ptr);

/* keepdata into HOST memory: host_traj is n_paths x (n_periods + 1) floats path-major,
 * host_final (may be NULL) n_paths floats.  Produced in device-sized slices.
 * Synchronous.  mc_simulations_keepdata, src/simulations.cpp:139-202. */
int smmc_engine_simulate_keepdata_to_host(smmc_engine *e, const smmc_sim *sim, float *host_traj,
                                          float *host_final);

/* ---- statistics of values already in HBM ---------------------------- */

#define SMMC_MAX_RANKS 8

/* One pass over n device floats -> packed statistics record (d_stats, device,
 * smmc_stats_bytes(n_bins) bytes): sum, sum of squares, count below a threshold, min,
 * max, bucket histogram.  Replaces the host passes update_mean_std and
 * update_count_below_min of the reference's callers (examples/visualize_returns_cpu_v2.cpp:
 * 113-138, examples/benchmark_mc_gpu.cpp:7-41) on data that never leaves the GPU.
 * Asynchronous on the engine stream. */
int smmc_engine_values_stats(smmc_engine *e, const float *d_values, uint64_t n, float below_threshold,
                             uint32_t n_bins, float hist_lo, float hist_hi, void *d_stats);

/* Exact order statistics: host_out[q] = the ranks[q]-th smallest (0-based) of n device
 * floats, n_ranks <= SMMC_MAX_RANKS, every rank < n.  Three histogram passes of radix
 * selection; the input is not modified or copied.  Synchronous. */
int smmc_engine_order_statistics(smmc_engine *e, const float *d_values, uint64_t n, const uint64_t *ranks,
                                 uint32_t n_ranks, float *host_out);

/* {min, Q1, Q2, Q3, max} with Q1 = n/4, Q2 = n/2, Q3 = Q1 + Q2 as ranks in sorted order:
 * update_quartiles, examples/visualize_returns_cpu_v2.cpp:83-111.  n >= 1.  Synchronous. */
int smmc_engine_quartiles(smmc_engine *e, const float *d_values, uint64_t n, float host_out[5]);

/* Mean of n HOST floats: chunked host-to-device copy + values_stats, double
 * accumulation; *mean = (float)sum / n as the CPU check of
 * examples/benchmark_reduce_mean.cpp:31-32.  reduce_mean_gpu, src/simulations.cu:269-341
 * (which sums in float by an in-place strided tree and overflows int indices beyond
 * 2^31 elements).  sum may be NULL.  Synchronous. */
int smmc_engine_reduce_mean_host(smmc_engine *e, const float *host_values, uint64_t n, float *mean, double *sum);

/* The same two operations on n HOST floats (copied to the device once, synchronous):
 * what update_quartiles / update_mean_std / update_count_below_min of the reference's
 * examples do on their std::vector.  stats and hist may be NULL, quartiles may be NULL. */
int smmc_engine_host_values_summary(smmc_engine *e, const float *host_values, uint64_t n, float below_threshold,
                                    uint32_t n_bins, float hist_lo, float hist_hi, smmc_stats *stats,
                                    uint64_t *hist, float quartiles[5]);

/* Device-time instrumentation: when enabled, every simulate call brackets its
 * main kernel with HIP events on the engine stream.  smmc_engine_kernel_ms
 * synchronises, returns the number of timed launches and their summed duration
 * in milliseconds, and clears the log. */
int smmc_engine_timing(smmc_engine *e, int enable);
int smmc_engine_kernel_ms(smmc_engine *e, double *total_ms, uint32_t *launches);
/* With timing enabled every workgroup of the path kernel also adds the shader clocks and the 100 MHz ticks of its
 * own lifetime to two counters: *ghz = the clock the chip HELD, on average over the workgroups of the launches since
 * the last call (0 when nothing was sampled; the reference-stream kernels are not sampled).  Synchronises and
 * clears.  The reference's counterpart is its printed phase timers (src/simulations.cu:351-358): it has no clock
 * read-out. */
int smmc_engine_kernel_clock(smmc_engine *e, double *ghz);

/* Device self-test of the kernels' divide shortcut over the binary32 bit patterns
 * [bits_lo, bits_hi): counts x where the reciprocal-multiply divide differs from the
 * IEEE x / 100.0f.  Synchronous. */
int smmc_engine_selftest(smmc_engine *e, uint32_t bits_lo, uint32_t bits_hi, uint64_t *div_mismatches);

/* Which divide-by-100 a launch of `sim` uses (the result never depends on it): SMMC_DIV_FAST, the
 * reciprocal-multiply form, when the returns table (or mean +- 7 std), the capital and the number of
 * periods prove that no path can leave its domain; SMMC_DIV_CHECKED (final-value launches only)
 * when they do not but a per-block range check with an IEEE-divide rerun of the rare offending path
 * is possible; SMMC_DIV_EXACT, the IEEE divide, otherwise or with SMMC_FLAG_EXACT_DIV.
 * keepdata != 0 asks for smmc_engine_simulate_keepdata.  Returns the kind (>= 0) or an error. */
#define SMMC_DIV_FAST 0
#define SMMC_DIV_EXACT 1
#define SMMC_DIV_CHECKED 2
int smmc_engine_divide_kind(smmc_engine *e, const smmc_sim *sim, int keepdata);

/* Launch geometry the engine will use (workgroups x threads), for reports. */
int smmc_engine_geometry(smmc_engine *e, uint32_t *grid, uint32_t *block, uint32_t *compute_units);

/* ---- several devices of one process ----------------------------------------------------- */

/* A group runs ONE simulation request sharded over several devices of this process and produces ONE
 * merged result: the multi-GPU launcher of the reference (mc_simulations_multi_gpu_launcher_async,
 * src/simulations.cu:576-655, reached through mc_simulations_gpu(..., n_gpus), :661-680) with the
 * defects SURVEY section 0.7 lists removed -- shard g covers floor(N/G) paths plus one of the N mod G
 * leftovers (the reference drops them, :602-603), global path ids are the stream counter so the
 * result does not depend on G (the reference replays the same seeds on every GPU, :120,140), one host
 * thread per device so that all devices compute and copy at once.
 *
 * How the per-device statistics records become one is chosen at creation:
 *   SMMC_MERGE_HOST  every device's record (864 bytes at 100 buckets) is already in host memory when its
 *                    thread returns; they are added in device order.
 *   SMMC_MERGE_RCCL  ncclCommInitAll over the group's devices (once, kept for the group's lifetime;
 *                    librccl.so.1 is opened only then), and per call ONE grouped all-reduce
 *                    (ncclUint64, ncclSum) over [count, below, underflow, overflow] and the bucket
 *                    counts, after which EVERY device holds the merged integer record in its own HBM
 *                    (smmc_group_device_record); the two double sums and min / max are merged on
 *                    the host in device order (an all-reduce would make them arrival-order
 *                    dependent).  Needs distinct devices.
 * Both give the same bits.  DESIGN.md section 7 has the measured cost of each. */
typedef struct smmc_group smmc_group;
#define SMMC_MERGE_HOST 0
#define SMMC_MERGE_RCCL 1

/* devices[0 .. n_devices): HIP device ids; with SMMC_MERGE_HOST a device may appear more than once
 * (each occurrence gets its own engine and stream: several shards on one GPU). */
int smmc_group_create(const int *devices, int n_devices, int merge, smmc_group **out);
void smmc_group_destroy(smmc_group *g);
int smmc_group_size(const smmc_group *g);

/* smmc_engine_set_table on every device of the group.  A table identical to the one the devices hold is not
 * uploaded again.  If a device fails, the devices may hold different tables: the group then refuses table-mode
 * simulations (SMMC_ERR_INVALID) until a smmc_group_set_table has succeeded on all of them. */
int smmc_group_set_table(smmc_group *g, const float *returns_percent, uint32_t n);

/* Progress of smmc_group_simulate, summed over the devices (see smmc_engine_set_progress). */
int smmc_group_set_progress(smmc_group *g, smmc_progress_fn fn, void *user);

/* smmc_engine_simulate_to_host for the whole request: device g simulates its contiguous share of the
 * global path ids sim->first_path .. first_path + n_paths - 1 and streams it to its place in the host
 * arrays (any may be NULL; a pageable host_final of 32 MiB or more is page-locked ONCE for all devices, by the
 * engine's own rules -- SMMC_PIN_HOST, both ends tested for "pinned already" -- and the shards run with
 * SMMC_FLAG_HOST_NOPIN, also when that registration fails);
 * stats / hist receive the merged record.  The chunk arrays need every shard to start on a multiple
 * of SMMC_CHUNK paths: SMMC_ERR_INVALID otherwise.  Synchronous. */
int smmc_group_simulate(smmc_group *g, const smmc_sim *sim, float *host_final, float *host_chunk_mean,
                        float *host_chunk_var, volatile int64_t *progress, smmc_stats *stats, uint64_t *hist);

/* smmc_engine_prepare_host on every device of the group, for its share of n_paths (in parallel). */
int smmc_group_prepare_host(smmc_group *g, uint64_t n_paths);

/* The shard device `index` of the group gets of an n_paths request: its first path (relative to
 * sim->first_path) and its count. */
int smmc_group_shard(const smmc_group *g, uint64_t n_paths, int index, uint64_t *first, uint64_t *count);

/* SMMC_MERGE_RCCL, after a smmc_group_simulate that asked for statistics: the device pointer (on device
 * `index` of the group) of that device's copy of the merged packed record -- integer fields and
 * bucket counts are the merged ones, sum / sumsq / min / max that device's own. */
int smmc_group_device_record(smmc_group *g, int index, void **d_record);

/* Host wall-clock costs in milliseconds, for reports: creating the engines, creating the communicator
 * (opening librccl the first time included; 0 for SMMC_MERGE_HOST), and the merge step of the last
 * smmc_group_simulate. */
int smmc_group_timings(const smmc_group *g, double *engines_ms, double *comm_init_ms, double *last_merge_ms);

/* ---- statistics record helpers (host) --------------------------------------- */

uint64_t smmc_stats_bytes(uint32_t n_bins);
/* dst += src for two packed records with equal n_bins (host memory).  Merging
 * shards in ascending rank order gives a result independent of timing. */
int smmc_stats_merge(void *dst_packed, const void *src_packed);

/* ---- the reference's device demo ------------------------------------------------ */

/* vector_add_gpu, src/gpu.cu:17-47 (kernel impl_vector_add_gpu :8-14): out[i] = a[i] + b[i] for three HOST
 * arrays of n floats, computed on the current device (H2D of a and b, one launch, D2H of out; device
 * memory is allocated and freed inside the call, as in the reference).  north_star names the file beside
 * the engine; it is not part of the hot path.  kernel_seconds (optional): the launch alone, by HIP events
 * -- the figure the reference prints as "GPU time".  n == 0 is a no-op; SMMC_ERR_NO_DEVICE without a GPU. */
int smmc_vector_add(float *out, const float *a, const float *b, int64_t n, double *kernel_seconds);

#ifdef __cplusplus
}
#endif
#endif /* SMMC_H */
