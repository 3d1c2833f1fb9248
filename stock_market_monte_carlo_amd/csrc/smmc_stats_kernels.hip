// smmc_stats_kernels.hip -- distribution statistics of an array of final values that
// already lives in HBM.  These are the callers' next step on the path's output
//  and, unlike the simulation kernels, genuinely HBM-bound:
// 4 bytes read per value and nothing written.
//
//   values_stats_kernel  sum / sum of squares (double), count below a threshold, min, max
//                        and a bucket histogram in ONE pass: the host passes
//                        update_mean_std + update_count_below_min of the reference's
//                        examples (visualize_returns_cpu_v2.cpp:113-138,
//                        benchmark_mc_gpu.cpp:7-41) and the accumulation of
//                        reduce_mean_gpu (src/simulations.cu:269-341, which sums in
//                        float through a strided in-place tree; here double).
//   radix_hist_kernel +  exact order statistics (k-th smallest) by 3-pass MSD radix
//   radix_pick_kernel    selection on the order-preserving key of the binary32 pattern,
//                        11 + 11 + 10 bits: update_quartiles (visualize_returns_cpu_v2.cpp:
//                        83-111 uses three std::nth_element calls, visualize_returns_gpu.cpp:
//                        83-110 a full std::sort) without sorting or copying anything.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "smmc_internal.h"

namespace smmc {
namespace {

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
  return v;
}


// Streams the float4 body of an array through f, grid-wide: two ADJACENT 16-byte loads in flight
// per lane (a wave reads 2 KiB contiguous).  Measured on 1e9 values against the alternatives
// (A/B builds of this loop on one box): two loads a grid apart 5.0-5.5 TB/s, this 5.4-5.7, nontemporal
// loads the same within 2 %, one load per iteration 3.4-5.7, four adjacent loads 3.7-3.9; a bare
// read-and-add kernel reaches 6.2-6.4 TB/s on this chip (tools/ubench_read.hip).
template <typename F>
__device__ __forceinline__ void stream_float4(const float4 *body, uint64_t n4, uint64_t gtid, uint64_t gsize, F &&f) {
  const uint64_t n8 = n4 >> 1;
  for (uint64_t j = gtid; j < n8; j += gsize) {
    const float4 q0 = body[2 * j], q1 = body[2 * j + 1];
    f(q0);
    f(q1);
  }
  if ((n4 & 1u) && gtid == 0) f(body[n4 - 1]);
}

struct Acc {
  double sum = 0.0, sumsq = 0.0;
  uint32_t count = 0, below = 0, under = 0, over = 0;  // a lane sees < 2^32 values per launch
  float vmin = __builtin_inff(), vmax = -__builtin_inff();
};

// LDS histograms are replicated: lane l counts into copy l % copies, and copies are an odd
// number of words apart, so lanes that hit the same bucket (final values crowd into a few
// buckets) neither serialise on one address nor collide on one bank.
__device__ __forceinline__ uint32_t hist_stride(uint32_t bins) { return bins | 1u; }

template <bool kHist>
__device__ __forceinline__ void take(Acc &a, float v, const ValuesArgs &k, uint32_t *lds_hist) {
  const double dv = static_cast<double>(v);
  a.sum += dv;
  a.sumsq += dv * dv;  // dv * dv is exact in binary64
  a.count += 1;
  a.below += (v < k.below_threshold) ? 1u : 0u;
  a.vmin = fminf(a.vmin, v);
  a.vmax = fmaxf(a.vmax, v);
  if constexpr (kHist) {
    if (v < k.hist_lo) {
      a.under += 1;
    } else if (v < k.hist_hi) {
      int32_t b = static_cast<int32_t>((dv - static_cast<double>(k.hist_lo)) * k.hist_inv);
      b = b < static_cast<int32_t>(k.n_bins) - 1 ? b : static_cast<int32_t>(k.n_bins) - 1;
      atomicAdd(&lds_hist[b], 1u);
    } else {
      a.over += 1;
    }
  }
}

// One pass over n floats: 16-byte loads on the aligned body, scalar head and tail.
// 1024-thread workgroups, two per CU: every workgroup is resident for the whole launch and flushes
// its histogram at the same moment at the end, so the number of workgroups sets the length of that
// tail (2048 workgroups of 256 threads: 0.099 ms for 1e8 values; spread over 16 bucket arrays: 0.086;
// 512 of 1024 threads: see DESIGN.md section 5).
constexpr int kStatsBlock = 1024;
constexpr int kStatsWaves = kStatsBlock / 64;
template <bool kHist>
__global__ __launch_bounds__(kStatsBlock) void values_stats_kernel(const ValuesArgs k) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  uint32_t *lds_all = reinterpret_cast<uint32_t *>(lds_raw);
  __shared__ BlockPartial wave_part[kStatsWaves];
  const uint32_t tid = threadIdx.x;
  const uint32_t stride = hist_stride(k.n_bins);
  uint32_t *lds_hist = lds_all + (tid % k.hist_copies) * stride;  // this lane's copy
  if constexpr (kHist) {
    for (uint32_t i = tid; i < k.hist_copies * stride; i += kStatsBlock) lds_all[i] = 0u;
    __syncthreads();
  }
  Acc a;
  const uint64_t gtid = static_cast<uint64_t>(blockIdx.x) * kStatsBlock + tid;
  const uint64_t gsize = static_cast<uint64_t>(gridDim.x) * kStatsBlock;
  // head: elements before the first 16-byte boundary
  const uint64_t mis = (reinterpret_cast<uintptr_t>(k.values) >> 2) & 3u;
  uint64_t head = mis ? 4 - mis : 0;
  if (head > k.n) head = k.n;
  if (gtid < head) take<kHist>(a, k.values[gtid], k, lds_hist);
  const float4 *body = reinterpret_cast<const float4 *>(k.values + head);
  const uint64_t n4 = (k.n - head) >> 2;
  stream_float4(body, n4, gtid, gsize, [&](const float4 &q) {
    take<kHist>(a, q.x, k, lds_hist);
    take<kHist>(a, q.y, k, lds_hist);
    take<kHist>(a, q.z, k, lds_hist);
    take<kHist>(a, q.w, k, lds_hist);
  });
  const uint64_t tail0 = head + (n4 << 2);
  if (tail0 + gtid < k.n) take<kHist>(a, k.values[tail0 + gtid], k, lds_hist);  // < 4 elements

  BlockPartial p;
  p.sum = wave_sum(a.sum);
  p.sumsq = wave_sum(a.sumsq);
  p.count = wave_sum(static_cast<unsigned long long>(a.count));
  p.below = wave_sum(static_cast<unsigned long long>(a.below));
  p.underflow = wave_sum(static_cast<unsigned long long>(a.under));
  p.overflow = wave_sum(static_cast<unsigned long long>(a.over));
  p.min = wave_min(a.vmin);
  p.max = wave_max(a.vmax);
  const int lane = tid & 63, wave = tid >> 6;
  if (lane == 0) wave_part[wave] = p;
  __syncthreads();
  if (tid == 0) {
    BlockPartial t = wave_part[0];
#pragma unroll
    for (int w = 1; w < kStatsWaves; ++w) {
      t.sum += wave_part[w].sum;
      t.sumsq += wave_part[w].sumsq;
      t.count += wave_part[w].count;
      t.below += wave_part[w].below;
      t.underflow += wave_part[w].underflow;
      t.overflow += wave_part[w].overflow;
      t.min = fminf(t.min, wave_part[w].min);
      t.max = fmaxf(t.max, wave_part[w].max);
    }
    k.partials[blockIdx.x] = t;
  }
  if constexpr (kHist) {  // the barrier above also closed the LDS atomics
    for (uint32_t b = tid; b < k.n_bins; b += kStatsBlock) {
      uint32_t c = 0;
      for (uint32_t r = 0; r < k.hist_copies; ++r) c += lds_all[r * stride + b];
      unsigned long long *dst = k.spread ? k.hist_spread + static_cast<size_t>(blockIdx.x % k.spread) * k.n_bins : k.d_hist;
      if (c) atomicAdd(&dst[b], static_cast<unsigned long long>(c));
    }
  }
}

// ---- exact order statistics: MSD radix selection ---------------------------------------

// Order-preserving map binary32 -> u32 (negative values reversed below positive ones).
__device__ __forceinline__ uint32_t order_key(float v) {
  const uint32_t b = __float_as_uint(v);
  return b ^ ((b & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t key) {
  return __uint_as_float(key ^ ((key & 0x80000000u) ? 0x80000000u : 0xFFFFFFFFu));
}

constexpr uint32_t kRadixBins = 2048;
constexpr uint32_t kRadixCopies = 7;  // 7 x 2049 words = 56 KiB

constexpr int kRadixBlock = 1024;  // 16 waves share one set of LDS histograms: 2 workgroups fill a CU

// Pass 0: every value counts, by its top 11 key bits.  Passes 1 / 2: a value counts into the histogram of the one
// group whose prefix it shares (if any), by its next 11 / last 10 bits: one (predicated) LDS atomic per value whatever
// the number of groups.
//
// Finding the group.  Round 2's select chain -- for each of the kMaxRanks possible groups a compare and a select, both
// half rate -- cost 64 of the ~80 clocks a value took, and passes 1 / 2 ran at 82-87 us per 1e8 values where pass 0
// (7 VALU per value) takes 67.  Now the 11 key bits that tell this pass's groups apart -- the top 11 in pass 1, the
// middle 11 in pass 2 -- index a 2048-entry LDS table whose entry is (the group's whole prefix << 3 | group), or
// all ones: one ds_read_b32, a shift and ONE compare say whether the value belongs to a group and to which.  Groups
// have distinct prefixes by construction (radix_pick_kernel), so in pass 1 their table slots are distinct; in pass 2
// two groups can share their middle 11 bits while differing above them (1.5 and 3.0 do): the workgroup that finds
// such a collision while filling the table takes the select chain instead (kernel-uniform: every workgroup sees the
// same groups).  Both forms count exactly the same values.
constexpr uint32_t kNoGroup = 0xFFFFFFFFu;

template <int kPass>
__device__ __forceinline__ void radix_take_chain(uint32_t key, uint32_t n_groups, const uint32_t *gprefix, uint32_t *lds_hist) {
  constexpr uint32_t kShift = kPass == 1 ? 21u : 10u;
  const uint32_t head = key >> kShift;
  int32_t slot = -1;
#pragma unroll
  for (uint32_t g = 0; g < kMaxRanks; ++g)
    slot = (g < n_groups && head == gprefix[g]) ? static_cast<int32_t>(g * kRadixBins) : slot;
  if (slot >= 0) atomicAdd(&lds_hist[slot + (kPass == 1 ? ((key >> 10) & 2047u) : (key & 1023u))], 1u);
}

template <int kPass>
__device__ __forceinline__ void radix_take_table(uint32_t key, const uint32_t *lds_tab, uint32_t *lds_hist) {
  constexpr uint32_t kShift = kPass == 1 ? 21u : 10u;
  const uint32_t head = key >> kShift;                          // 11 or 22 bits
  const uint32_t e = lds_tab[kPass == 1 ? head : (head & 2047u)];
  if ((e >> 3) == head)                                         // kNoGroup >> 3 is no 22-bit head
    atomicAdd(&lds_hist[(e & 7u) * kRadixBins + (kPass == 1 ? ((key >> 10) & 2047u) : (key & 1023u))], 1u);
}
static_assert(kMaxRanks <= 8, "a table entry keeps the group in three bits");

template <int kPass, typename Take>
__device__ __forceinline__ void radix_stream(const float *values, uint64_t n, Take &&take_one) {
  const uint32_t tid = threadIdx.x;
  const uint64_t gtid = static_cast<uint64_t>(blockIdx.x) * kRadixBlock + tid;
  const uint64_t gsize = static_cast<uint64_t>(gridDim.x) * kRadixBlock;
  const uint64_t mis = (reinterpret_cast<uintptr_t>(values) >> 2) & 3u;
  uint64_t head = mis ? 4 - mis : 0;
  if (head > n) head = n;
  if (gtid < head) take_one(order_key(values[gtid]));
  const float4 *body = reinterpret_cast<const float4 *>(values + head);
  const uint64_t n4 = (n - head) >> 2;
  stream_float4(body, n4, gtid, gsize, [&](const float4 &v) {
    take_one(order_key(v.x));
    take_one(order_key(v.y));
    take_one(order_key(v.z));
    take_one(order_key(v.w));
  });
  const uint64_t tail0 = head + (n4 << 2);
  if (tail0 + gtid < n) take_one(order_key(values[tail0 + gtid]));
}

// LDS: pass 0 kRadixCopies histograms; passes 1 / 2 one histogram per possible group (n_ranks: the group count lives on
// the device), then the 2048-entry group table and one flag word.
template <int kPass>
__global__ __launch_bounds__(kRadixBlock) void radix_hist_kernel(const float *values, uint64_t n,
                                                                 const SelectState *st, unsigned long long *g_hist,
                                                                 uint32_t n_ranks, uint32_t force_chain) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  uint32_t *lds_all = reinterpret_cast<uint32_t *>(lds_raw);
  const uint32_t n_groups = kPass == 0 ? 1u : st->n_groups;
  const uint32_t tid = threadIdx.x;
  // pass 0 sees every value and they crowd into a few exponent bins: kRadixCopies lane-
  // interleaved copies (see hist_stride); later passes spread over mantissa bits, one copy
  const uint32_t words = kPass == 0 ? kRadixCopies * hist_stride(kRadixBins) : n_groups * kRadixBins;
  uint32_t *lds_hist = kPass == 0 ? lds_all + (tid % kRadixCopies) * hist_stride(kRadixBins) : lds_all;
  for (uint32_t i = tid; i < words; i += kRadixBlock) lds_all[i] = 0u;
  if constexpr (kPass == 0) {
    __syncthreads();
    radix_stream<0>(values, n, [&](uint32_t key) { atomicAdd(&lds_hist[key >> 21], 1u); });
  } else {
    constexpr uint32_t kShift = kPass == 1 ? 21u : 10u;
    uint32_t *lds_tab = lds_all + n_ranks * kRadixBins, *lds_flag = lds_tab + kRadixBins;
    for (uint32_t i = tid; i < kRadixBins; i += kRadixBlock) lds_tab[i] = kNoGroup;
    uint32_t gprefix[kMaxRanks];  // wave-uniform: scalar registers
#pragma unroll
    for (uint32_t g = 0; g < kMaxRanks; ++g) gprefix[g] = st->group_prefix[g] >> kShift;
    __syncthreads();
    if (tid == 0) {
      uint32_t collide = force_chain;
      for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t head = st->group_prefix[g] >> kShift, slot = kPass == 1 ? head : (head & 2047u);
        if (lds_tab[slot] != kNoGroup) collide = 1u;
        lds_tab[slot] = (head << 3) | g;
      }
      *lds_flag = collide;
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(*lds_flag))  // the same for every workgroup of the launch
      radix_stream<kPass>(values, n, [&](uint32_t key) { radix_take_chain<kPass>(key, n_groups, gprefix, lds_hist); });
    else
      radix_stream<kPass>(values, n, [&](uint32_t key) { radix_take_table<kPass>(key, lds_tab, lds_hist); });
  }
  __syncthreads();
  for (uint32_t b = tid; b < n_groups * kRadixBins; b += kRadixBlock) {
    uint32_t c = 0;
    if constexpr (kPass == 0) {
      for (uint32_t r = 0; r < kRadixCopies; ++r) c += lds_all[r * hist_stride(kRadixBins) + b];
    } else {
      c = lds_all[b];
    }
    if (c) atomicAdd(&g_hist[b], static_cast<unsigned long long>(c));
  }
}

// One wave per rank: finds the bin that holds the rank (counting within the rank's
// group histogram), appends the bin's bits to the prefix and makes the rank relative to
// the bin; then thread 0 regroups the ranks by their new prefixes.  After pass 2 the
// prefix is the whole key; out[q] receives the value.
__global__ __launch_bounds__(64 * kMaxRanks) void radix_pick_kernel(int pass, uint32_t n_ranks, SelectState *st,
                                                                    unsigned long long *g_hist, float *out) {
  const uint32_t groups_counted = pass == 0 ? 1u : st->n_groups;  // what radix_hist_kernel of this pass added into
  const uint32_t q = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (q < n_ranks) {
    const unsigned long long *h = g_hist + (pass == 0 ? 0u : st->group_of[q] * kRadixBins);
    const uint32_t bins = pass == 2 ? 1024u : kRadixBins;
    const unsigned long long rank = st->rank[q];
    // each lane sums a contiguous slice of bins, an inclusive wave scan orders the slices
    const uint32_t per = bins / 64;
    unsigned long long mine = 0;
    for (uint32_t b = 0; b < per; ++b) mine += h[lane * per + b];
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long up = __shfl_up(incl, off, 64);
      if (static_cast<int>(lane) >= off) incl += up;
    }
    const unsigned long long excl = incl - mine;
    if (rank >= excl && rank < incl) {  // exactly one lane: rank < total was checked by the host
      unsigned long long cum = excl;
      uint32_t b = lane * per;
      for (; b < lane * per + per - 1; ++b) {
        if (rank < cum + h[b]) break;
        cum += h[b];
      }
      const uint32_t shift = pass == 0 ? 21u : pass == 1 ? 10u : 0u;
      const uint32_t prefix = st->prefix[q] | (b << shift);
      st->prefix[q] = prefix;
      st->rank[q] = rank - cum;
      if (pass == 2) out[q] = key_value(prefix);
    }
  }
  __syncthreads();
  // every rank has read its histogram: leave the array ZERO for the next pass (and the next call), so that no pass is
  // preceded by a memset (round 4: three fewer launches per call)
  for (uint32_t b = threadIdx.x; b < groups_counted * kRadixBins; b += blockDim.x) g_hist[b] = 0ull;
  if (threadIdx.x == 0 && pass < 2) {
    uint32_t n_groups = 0;
    for (uint32_t r = 0; r < n_ranks; ++r) {
      const uint32_t p = st->prefix[r];
      uint32_t g = 0;
      while (g < n_groups && st->group_prefix[g] != p) ++g;
      if (g == n_groups) st->group_prefix[n_groups++] = p;
      st->group_of[r] = g;
    }
    st->n_groups = n_groups;
  }
}

}  // namespace

uint32_t values_hist_copies(uint32_t n_bins) {
  if (!n_bins) return 1;
  const uint32_t fit = (32u * 1024u / 4u) / (n_bins | 1u);  // keep the histogram within 32 KiB
  uint32_t most = 16;
  if (const char *env = std::getenv("SMMC_STATS_HIST_COPIES")) {  // tuning knob: 1 ... 64
    const long v = std::strtol(env, nullptr, 10);
    if (v >= 1 && v <= 64) most = static_cast<uint32_t>(v);
  }
  return fit >= most ? most : fit >= 1 ? fit : 1u;
}

hipError_t launch_values_stats(const ValuesArgs &a, uint32_t grid, hipStream_t stream) {
  if (a.n_bins)
    hipLaunchKernelGGL((values_stats_kernel<true>), dim3(grid), dim3(kStatsBlock),
                       a.hist_copies * (a.n_bins | 1u) * 4u, stream, a);
  else
    hipLaunchKernelGGL((values_stats_kernel<false>), dim3(grid), dim3(kStatsBlock), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_radix_hist(const float *values, uint64_t n, int pass, uint32_t n_ranks, const SelectState *st,
                             unsigned long long *g_hist, uint32_t grid, hipStream_t stream) {
  // sized for the worst case of one group per rank (the group count lives on the device); passes 1 / 2: + the group
  // table and its flag word
  const size_t lds = pass == 0 ? static_cast<size_t>(kRadixCopies) * (kRadixBins | 1u) * 4u
                               : (static_cast<size_t>(n_ranks) * kRadixBins + kRadixBins + 4u) * 4u;
  // SMMC_RADIX_MATCH=chain: the select chain of round 2 for every value (A/B runs and tests)
  static const uint32_t force_chain = [] {
    const char *env = std::getenv("SMMC_RADIX_MATCH");
    return (env && !std::strcmp(env, "chain")) ? 1u : 0u;
  }();
  if (lds > 60u * 1024u) {  // more than the default limit of dynamic LDS: opt in (160 KiB per CU on CDNA4; two of these workgroups fit)
    const void *fn = pass == 1 ? reinterpret_cast<const void *>(radix_hist_kernel<1>) : reinterpret_cast<const void *>(radix_hist_kernel<2>);
    const hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    if (err != hipSuccess) return err;
  }
  if (pass == 0)
    hipLaunchKernelGGL(radix_hist_kernel<0>, dim3(grid), dim3(kRadixBlock), lds, stream, values, n, st, g_hist, n_ranks, force_chain);
  else if (pass == 1)
    hipLaunchKernelGGL(radix_hist_kernel<1>, dim3(grid), dim3(kRadixBlock), lds, stream, values, n, st, g_hist, n_ranks, force_chain);
  else
    hipLaunchKernelGGL(radix_hist_kernel<2>, dim3(grid), dim3(kRadixBlock), lds, stream, values, n, st, g_hist, n_ranks, force_chain);
  return hipGetLastError();
}

hipError_t launch_radix_pick(int pass, uint32_t n_ranks, SelectState *st, unsigned long long *g_hist,
                             float *d_out, hipStream_t stream) {
  hipLaunchKernelGGL(radix_pick_kernel, dim3(1), dim3(64 * kMaxRanks), 0, stream, pass, n_ranks, st, g_hist, d_out);
  return hipGetLastError();
}

}  // namespace smmc
