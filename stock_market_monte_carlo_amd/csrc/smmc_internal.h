// smmc_internal.h -- shared between the kernel TU and the C-ABI TU (not installed).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "smmc.h"

namespace smmc {

constexpr int kBlock = SMMC_CHUNK;  // threads per workgroup = paths per chunk (4 waves)

// One workgroup's partial statistics; reduced in a fixed order by the finalize kernel.
struct BlockPartial {
  double sum, sumsq;
  unsigned long long count, below, underflow, overflow;
  float min, max;
};

struct KernelArgs {
  int32_t mode;           // SMMC_MODE_*
  const float *table_a;  // device, table_len entries, already 100.0f + r  (MODE_TABLE)
  const float *bm_tables;  // device, Box-Muller radius + trig tables of the launch's stream (MODE_GAUSSIAN), 16-byte aligned
  uint32_t table_len;
  uint32_t key0, key1;   // Philox key = seed lo, hi
  uint64_t first_path;
  uint64_t n_paths;
  uint32_t n_periods;
  float initial_capital;
  float gauss_mean, gauss_std;
  float gauss_shift100;  // 100.0f + gauss_mean: the additive term of counter stream v3's multiplier draw
  int32_t stream;        // Gaussian draw: 2 = counter stream v2 (SMMC_FLAG_STREAM_V2), else v3
  float *d_final;        // nullable
  float *d_chunk_mean;   // nullable
  float *d_chunk_var;    // nullable
  BlockPartial *partials;        // nullable => no statistics
  unsigned long long *d_hist;    // nullable; n_bins counters, pre-zeroed, in the packed record
  uint32_t n_bins;
  float hist_lo, hist_hi;
  double hist_inv;       // (double)n_bins / ((double)hi - (double)lo), computed on the host
  float below_threshold;
  float *d_traj;         // keepdata only: n_paths x (n_periods + 1), path-major
  float chk_lo, chk_hi;  // SMMC_DIV_CHECKED only: the window a path must stay in at Philox-block boundaries
  // Timing instrumentation (smmc_engine_timing): every workgroup of paths_kernel adds the shader clocks
  // (s_memtime) and the 100 MHz ticks (s_memrealtime) of its own lifetime -- their ratio is the clock the chip
  // HELD while this launch ran (smmc_engine_kernel_clock).  nullable.
  unsigned long long *clock_probe;
};

// The reference's own stream (SMMC_FLAG_STREAM_REF, smmc_ref_kernels.hip): per-path mt19937 seeded with
// seed0 + i, libstdc++'s Lemire map onto the table, update_fund -- src/simulations.cpp:240-252.
struct RefArgs {
  const float *table_a;   // device, table_len entries, 100.0f + r
  uint32_t table_len;
  uint32_t reject_below;  // Lemire: an output whose low product word is below (2^32 - T) % T is rejected
  uint32_t seed0;         // path i of the launch seeds its generator with (uint32_t)(seed0 + i)
  uint32_t n_paths;       // <= 2^31 per launch
  uint32_t n_periods;     // windowed kernel: <= ref_windowed_max_outputs()
  float initial_capital;
  float chk_lo, chk_hi;   // SMMC_DIV_CHECKED: as KernelArgs
  float *d_final;         // n_paths floats
  float *d_traj;          // nullable: n_paths rows of n_periods + 1 floats (keepdata)
  uint32_t traj_rows;     // d_traj, state-free kernels: consecutive rows per lane (8, 4, 2 or 1: TrajWriter, smmc_ref_kernels.hip)
  uint32_t *redo_count;   // windowed kernel: paths it left unfinished (appended to redo_list); generic kernel
  uint32_t *redo_list;    //   with redo_list != nullptr: the work items are redo_list[0 .. *redo_count)
  uint32_t *workspace;    // generic kernel: ref_workspace_bytes(grid)
};
uint32_t ref_windowed_max_outputs();
size_t ref_workspace_bytes(uint32_t grid);
size_t ref_windowed_lds_bytes(uint32_t table_len, bool traj);
hipError_t launch_ref_windowed(const RefArgs &a, int div, uint32_t grid, hipStream_t stream);
hipError_t launch_ref_generic(const RefArgs &a, bool exact_div, uint32_t grid, hipStream_t stream);
hipError_t launch_chunk_stats(const float *values, uint64_t n, float *d_mean, float *d_var, uint32_t grid,
                              hipStream_t stream);

// values_stats_kernel arguments (smmc_stats_kernels.hip)
struct ValuesArgs {
  const float *values;
  uint64_t n;
  float below_threshold;
  uint32_t n_bins;
  uint32_t hist_copies;  // lane-interleaved LDS histogram copies (values_hist_copies)
  float hist_lo, hist_hi;
  double hist_inv;
  BlockPartial *partials;
  unsigned long long *d_hist;
  // values_stats: every workgroup is resident for the whole launch and flushes its histogram at the
  // same moment; `spread` zeroed copies of the bucket array (workgroup b adds to copy b % spread) cut
  // the adds that queue on one address from 2048 to 128; finalize_kernel folds the copies
  unsigned long long *hist_spread;
  uint32_t spread;
};
uint32_t values_hist_copies(uint32_t n_bins);
constexpr uint32_t kHistSpread = 16;
// Counter stream v3's Box-Muller tables (tools/gen_bm_tables.py; smmc_capi.cpp checks these against the
// generated smmc_bm_tables.inc): sub-intervals per radius octave, sectors of the angle table, and the two
// constants of the residual angle delta = fma(y, K, -C).
constexpr uint32_t kBm3SubBits = 3, kBm3TrigBits = 11, kBm3AngleBits = 30;
constexpr float kBm3AngleK = 0x1.921fb6p-5f, kBm3AngleC = 0x1.9eb0b4p-5f;

// radix selection state: per requested rank, the key bits fixed so far and the rank
// relative to the values that share those bits
constexpr uint32_t kMaxRanks = 8;
struct SelectState {
  uint32_t prefix[kMaxRanks];
  unsigned long long rank[kMaxRanks];
  // ranks that share a prefix share one histogram ("group"): a value then costs at most
  // one LDS atomic per pass however many ranks were asked for
  uint32_t n_groups;
  uint32_t group_prefix[kMaxRanks];
  uint32_t group_of[kMaxRanks];
};

hipError_t launch_values_stats(const ValuesArgs &a, uint32_t grid, hipStream_t stream);
hipError_t launch_radix_hist(const float *values, uint64_t n, int pass, uint32_t n_ranks, const SelectState *st,
                             unsigned long long *g_hist, uint32_t grid, hipStream_t stream);
// also zeroes the part of g_hist the pass counted into (the array is zero between passes and calls)
hipError_t launch_radix_pick(int pass, uint32_t n_ranks, SelectState *st, unsigned long long *g_hist,
                             float *d_out, hipStream_t stream);

// Launch wrappers (defined in smmc_kernels.hip).  All asynchronous on `stream`.
// `div`: SMMC_DIV_* of smmc.h (how a launch divides by 100: simulate_path in smmc_kernels.hip)
hipError_t launch_paths(const KernelArgs &a, int div, uint32_t grid, size_t lds_bytes,
                        hipStream_t stream);
// hist_acc: `spread` copies of n_bins bucket counts accumulated by the launch before; folded into the record and zeroed
// again (the engine keeps the array zero between launches: smmc_capi.cpp, hist_acc_ready)
hipError_t launch_finalize(const BlockPartial *partials, uint32_t n_partials, smmc_stats *d_stats,
                           uint32_t n_bins, hipStream_t stream, unsigned long long *hist_acc, uint32_t spread);
hipError_t launch_keepdata(const KernelArgs &a, bool exact_div, int tile, int waves, uint32_t grid,
                           hipStream_t stream);
// comb form of keepdata (rows [0, 2048 n_super) of a call; see keepdata_comb_kernel)
hipError_t launch_keepdata_comb(const KernelArgs &a, bool exact_div, int blocks_per_step, uint32_t rows_per_stream,
                                uint64_t n_wave_chunks, uint64_t n_rows_total, int waves, uint32_t grid,
                                unsigned long long *next_chunk, hipStream_t stream);
hipError_t launch_final_column(const float *traj, uint64_t n_rows, uint32_t row_len, float *d_final, uint32_t grid,
                               hipStream_t stream);
size_t keepdata_comb_lds_bytes(uint32_t table_len, int waves, int stream);
uint32_t keepdata_draws(uint32_t table_len);
hipError_t launch_selftest(uint32_t lo, uint32_t hi, unsigned long long *d_count, uint32_t grid,
                           hipStream_t stream);
size_t paths_lds_bytes(uint32_t table_len, uint32_t n_bins, int stream);
size_t keepdata_lds_bytes(uint32_t table_len, int tile, int waves, int stream);
size_t bm_tables_bytes(int stream);  // 2 | 3
hipError_t static_lds_bytes(size_t *bytes);  // of the kernels that address the v3 tables absolutely: 0

}  // namespace smmc
