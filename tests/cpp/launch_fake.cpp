// launch_fake.cpp -- with tests/cpp/fake_hip.cpp: stands in for the kernel translation units in CPU tests of the
// host-side code with several fake devices (TEST INFRASTRUCTURE; never part of the product, never a fallback:
// it does not simulate anything).  A "launch" of paths_kernel writes, on the host and at once, a KNOWN FUNCTION
// OF THE GLOBAL PATH ID (fake_path_value) where the real kernel writes a path's final value, and forms the chunk
// means / variances and the statistics record from those values with the kernel's own definitions -- so a test
// can check that every id of a sharded, chunked, multi-threaded run landed in its place exactly once and that
// the merged record is the record of all of them.  Every other kernel reports "no device".
#include <cmath>
#include <cstring>
#include <limits>

#include "smmc_internal.h"

extern "C" float fake_path_value(uint64_t id, uint32_t key0, uint32_t key1, uint32_t n_periods, float capital) {
  uint64_t z = id + 0x9E3779B97F4A7C15ull * (1 + key0) + (static_cast<uint64_t>(key1) << 32) + n_periods;  // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return capital * (0.5f + static_cast<float>(z & 0xFFFFu) / 65536.0f * 1.5f);
}

// test hook: the next n launch_finalize calls fail (an enqueue error between a kernel and its fold)
static int g_fail_finalize = 0;
extern "C" void fake_launch_fail_finalize(int n) { g_fail_finalize = n; }

namespace smmc {

hipError_t launch_paths(const KernelArgs &a, int, uint32_t grid, size_t, hipStream_t) {
  BlockPartial t;
  t.sum = t.sumsq = 0.0;
  t.count = t.below = t.underflow = t.overflow = 0;
  t.min = std::numeric_limits<float>::infinity();
  t.max = -t.min;
  double c1 = 0.0, c2 = 0.0;
  for (uint64_t i = 0; i < a.n_paths; ++i) {
    const float v = fake_path_value(a.first_path + i, a.key0, a.key1, a.n_periods, a.initial_capital);
    if (a.d_final) a.d_final[i] = v;
    const double dv = v;
    c1 += dv;
    c2 += dv * dv;
    if (i % kBlock == kBlock - 1 || i + 1 == a.n_paths) {  // one (mean, variance) pair per 256 paths, as paths_kernel
      const double n_in = static_cast<double>(i % kBlock + 1), mean = c1 / n_in, var = c2 / n_in - mean * mean;
      if (a.d_chunk_mean) a.d_chunk_mean[i / kBlock] = static_cast<float>(mean);
      if (a.d_chunk_var) a.d_chunk_var[i / kBlock] = static_cast<float>(var > 0.0 ? var : 0.0);
      c1 = c2 = 0.0;
    }
    if (!a.partials) continue;
    t.sum += dv;
    t.sumsq += dv * dv;
    t.count += 1;
    t.below += v < a.below_threshold ? 1 : 0;
    t.min = std::fmin(t.min, v);
    t.max = std::fmax(t.max, v);
    if (a.n_bins) {
      if (v < a.hist_lo) {
        t.underflow += 1;
      } else if (v < a.hist_hi) {
        int32_t b = static_cast<int32_t>((dv - static_cast<double>(a.hist_lo)) * a.hist_inv);
        b = b < static_cast<int32_t>(a.n_bins) - 1 ? b : static_cast<int32_t>(a.n_bins) - 1;
        a.d_hist[b] += 1;
      } else {
        t.overflow += 1;
      }
    }
  }
  if (a.partials) {
    BlockPartial none = t;
    none.sum = none.sumsq = 0.0;
    none.count = none.below = none.underflow = none.overflow = 0;
    none.min = std::numeric_limits<float>::infinity();
    none.max = -none.min;
    for (uint32_t g = 0; g < grid; ++g) a.partials[g] = g == 0 ? t : none;
  }
  return hipSuccess;
}

hipError_t launch_finalize(const BlockPartial *partials, uint32_t n_partials, smmc_stats *out, uint32_t n_bins, hipStream_t,
                           unsigned long long *hist_acc, uint32_t spread) {
  if (g_fail_finalize > 0) {
    --g_fail_finalize;
    return hipErrorLaunchFailure;
  }
  // as finalize_kernel: the bucket counts were accumulated in the engine's own array; fold, and leave it zero
  unsigned long long *hist = reinterpret_cast<unsigned long long *>(out + 1);
  for (uint32_t b = 0; spread && b < n_bins; ++b) {
    unsigned long long c = 0;
    for (uint32_t r = 0; r < spread; ++r) {
      c += hist_acc[static_cast<size_t>(r) * n_bins + b];
      hist_acc[static_cast<size_t>(r) * n_bins + b] = 0;
    }
    hist[b] = c;
  }
  out->count = out->below = out->underflow = out->overflow = 0;
  out->sum = out->sumsq = 0.0;
  out->min = std::numeric_limits<float>::infinity();
  out->max = -out->min;
  for (uint32_t j = 0; j < n_partials; ++j) {
    out->count += partials[j].count; out->below += partials[j].below;
    out->underflow += partials[j].underflow; out->overflow += partials[j].overflow;
    out->sum += partials[j].sum; out->sumsq += partials[j].sumsq;
    out->min = std::fmin(out->min, partials[j].min); out->max = std::fmax(out->max, partials[j].max);
  }
  out->n_bins = n_bins;
  out->reserved = 0;
  return hipSuccess;
}

size_t paths_lds_bytes(uint32_t table_len, uint32_t n_bins, int) { return (static_cast<size_t>(table_len) + n_bins) * 4u ; }
hipError_t static_lds_bytes(size_t *bytes) { *bytes = 0; return hipSuccess; }
size_t bm_tables_bytes(int stream) { return stream == 2 ? (1056 * 4 + 256 * 2) * 4 : (512 * 4 + 2048 * 2) * 4; }

// everything else: not part of what these tests drive
hipError_t launch_values_stats(const ValuesArgs &, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_radix_hist(const float *, uint64_t, int, uint32_t, const SelectState *, unsigned long long *, uint32_t,
                             hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_radix_pick(int, uint32_t, SelectState *, unsigned long long *, float *, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_keepdata(const KernelArgs &, bool, int, int, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_keepdata_comb(const KernelArgs &, bool, int, uint32_t, uint64_t, uint64_t, int, uint32_t, unsigned long long *,
                                hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_final_column(const float *, uint64_t, uint32_t, float *, uint32_t, hipStream_t) { return hipErrorNoDevice; }
size_t keepdata_comb_lds_bytes(uint32_t, int, int) { return 0; }
uint32_t keepdata_draws(uint32_t table_len) { return (table_len && table_len <= 2048u) ? 8u : 4u; }
hipError_t launch_selftest(uint32_t, uint32_t, unsigned long long *, uint32_t, hipStream_t) { return hipErrorNoDevice; }
uint32_t values_hist_copies(uint32_t) { return 1; }
size_t keepdata_lds_bytes(uint32_t, int, int, int) { return 0; }
uint32_t ref_windowed_max_outputs() { return 1816; }
size_t ref_workspace_bytes(uint32_t grid) { return static_cast<size_t>(grid) * 256 * 624 * 4; }
size_t ref_windowed_lds_bytes(uint32_t table_len, bool) { return static_cast<size_t>(table_len) * 4; }
hipError_t launch_ref_windowed(const RefArgs &, int, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_ref_generic(const RefArgs &, bool, uint32_t, hipStream_t) { return hipErrorNoDevice; }
hipError_t launch_chunk_stats(const float *, uint64_t, float *, float *, uint32_t, hipStream_t) { return hipErrorNoDevice; }
}  // namespace smmc

extern "C" int smmc_vector_add(float *, const float *, const float *, int64_t, double *) { return SMMC_ERR_NO_DEVICE; }
