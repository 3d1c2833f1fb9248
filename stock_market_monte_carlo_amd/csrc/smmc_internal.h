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
  const float *bm_tables;  // device, Box-Muller radius + trig tables (MODE_GAUSSIAN), 16-byte aligned
  uint32_t table_len;
  uint32_t key0, key1;   // Philox key = seed lo, hi
  uint64_t first_path;
  uint64_t n_paths;
  uint32_t n_periods;
  float initial_capital;
  float gauss_mean, gauss_std;
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
};

// Launch wrappers (defined in smmc_kernels.hip).  All asynchronous on `stream`.
hipError_t launch_paths(const KernelArgs &a, bool exact_div, uint32_t grid, size_t lds_bytes,
                        hipStream_t stream);
hipError_t launch_finalize(const BlockPartial *partials, uint32_t n_partials, smmc_stats *d_stats,
                           uint32_t n_bins, hipStream_t stream);
hipError_t launch_keepdata(const KernelArgs &a, bool exact_div, int tile, uint32_t grid, hipStream_t stream);
hipError_t launch_selftest(uint32_t lo, uint32_t hi, unsigned long long *d_count, uint32_t grid,
                           hipStream_t stream);
size_t paths_lds_bytes(uint32_t table_len, uint32_t n_bins);
size_t keepdata_lds_bytes(uint32_t table_len, int tile);
size_t bm_tables_bytes();

}  // namespace smmc
