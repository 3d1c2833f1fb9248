// smmc_kernels.hip -- gfx950 kernels of the Monte-Carlo returns engine.
//
// Replaces mc_simulations_gpu_kernel (reference src/simulations.cu:112-152) and
// mc_simulations_gpu_kernel_reduceBlock (:185-247); written from scratch for CDNA4
// (wave64, LDS-staged table, wave shuffles + LDS for reductions, no MFMA: the path
// has no contraction).  The arithmetic follows the CPU semantics
// (src/simulations.cpp:14-16, three binary32 roundings per period), not the
// reference GPU kernel's `total += total * (r * 0.01f)`.
//
// Bound: VALU issue (Philox4x32-10 + the dependent compounding chain).  HBM sees
// 4 B per path (the coalesced final-value store) or nothing (statistics only).
//
// Build with -ffp-contract=off: every fused multiply-add is an explicit
// __builtin_fmaf; results must be bit-identical to oracle/smmc_oracle.c engine (C).
#include <hip/hip_runtime.h>

#include "smmc_internal.h"

namespace smmc {
namespace {

constexpr uint32_t kPhiloxM0 = 0xD2511F53u;
constexpr uint32_t kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kWeyl0 = 0x9E3779B9u;
constexpr uint32_t kWeyl1 = 0xBB67AE85u;
constexpr uint32_t kDenseMaxTable = 2048u;  // largest table drawn eight-per-block

// a ^ b ^ c in one VALU instruction (gfx950 v_bitop3_b32, truth table 0x96); hipcc
// does not form it from two chained XORs on its own.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// Philox4x32-10 (Salmon et al., SC'11).  The key schedule is wave-uniform and
// lives in SGPRs; per round the lanes pay two 32x32->64 multiplies
// (v_mad_u64_u32) and two three-input XORs (v_bitop3_b32).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = static_cast<uint64_t>(kPhiloxM0) * c0;
    const uint64_t p1 = static_cast<uint64_t>(kPhiloxM1) * c2;
    const uint32_t n0 = xor3(static_cast<uint32_t>(p1 >> 32), c1, k0);
    const uint32_t n2 = xor3(static_cast<uint32_t>(p0 >> 32), c3, k1);
    c1 = static_cast<uint32_t>(p1);
    c3 = static_cast<uint32_t>(p0);
    c0 = n0;
    c2 = n2;
    k0 += kWeyl0;
    k1 += kWeyl1;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// x / 100.0f, correctly rounded.  The fast form (reciprocal multiply + one FMA
// residual correction) equals the IEEE quotient for every binary32 x with
// |x| >= 2^-124 (exhaustive CPU proof: tests/test_numerics_cpu.py); the host only
// selects it when the whole trajectory provably stays inside [2^-100, 2^100].
template <bool kExactDiv>
__device__ __forceinline__ float div100(float m) {
  if constexpr (kExactDiv) {
    return m / 100.0f;  // IEEE divide (hipcc default: correctly rounded)
  } else {
    const float c = 0.01f;
    const float q = m * c;
    const float e = __builtin_fmaf(-100.0f, q, m);
    return __builtin_fmaf(e, c, q);
  }
}

// One compounding period, src/simulations.cpp:14-16 with a = 100.0f + r formed
// by the caller: total * a, then / 100.
template <bool kExactDiv>
__device__ __forceinline__ float compound(float total, float a) {
  const float m = total * a;
  return div100<kExactDiv>(m);
}

// sqrt(x) correctly rounded, for x = +0 or x in [2^-40, 64] (no input scaling, no
// inf/NaN handling: the Box-Muller argument is 0 or lies in [1.19e-7, 45.8]).  v_sqrt_f32 is within
// 1 ulp; the two FMA residuals pick the neighbour when it is the better rounding.
// Bit-identical to the IEEE sqrtf on that domain (device self-test: smmc_selftest).
__device__ __forceinline__ float sqrt_rn_pos(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u);
  const float s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_dn = __builtin_fmaf(-s_dn, s, x);
  const float r_up = __builtin_fmaf(-s_up, s, x);
  s = (r_dn <= 0.0f) ? s_dn : s;
  s = (r_up > 0.0f) ? s_up : s;
  return s;
}

// ln(x), x a normal positive binary32: Cephes-style degree-8 kernel on
// [sqrt(1/2), sqrt(2)).  Same operation sequence as the oracle's log kernel.
__device__ __forceinline__ float log_kernel(float x) {
  uint32_t ix = __float_as_uint(x);
  ix += 0x3f800000u - 0x3f3504f3u;
  const int32_t e = static_cast<int32_t>(ix >> 23) - 127;
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  const float f = __uint_as_float(ix) - 1.0f;
  const float fe = static_cast<float>(e);
  const float z = f * f;
  float p = 7.0376836292E-2f;
  p = __builtin_fmaf(p, f, -1.1514610310E-1f);
  p = __builtin_fmaf(p, f, 1.1676998740E-1f);
  p = __builtin_fmaf(p, f, -1.2420140846E-1f);
  p = __builtin_fmaf(p, f, 1.4249322787E-1f);
  p = __builtin_fmaf(p, f, -1.6668057665E-1f);
  p = __builtin_fmaf(p, f, 2.0000714765E-1f);
  p = __builtin_fmaf(p, f, -2.4999993993E-1f);
  p = __builtin_fmaf(p, f, 3.3333331174E-1f);
  const float fz = f * z;
  float y = fz * p;
  y = __builtin_fmaf(fe, -2.12194440e-4f, y);
  y = __builtin_fmaf(-0.5f, z, y);
  float r = f + y;
  r = __builtin_fmaf(fe, 0.693359375f, r);
  return r;
}

// Box-Muller on two 32-bit words (ua: radius, ub: angle); see DESIGN.md section 3.
__device__ __forceinline__ void box_muller(uint32_t ua, uint32_t ub, float &z_cos, float &z_sin) {
  const float u1 = __builtin_fmaf(static_cast<float>(ua), 0x1p-32f, 0x1p-33f);  // (0, 1]
  const float t = -2.0f * log_kernel(u1);
  const float r = sqrt_rn_pos(t);  // == IEEE sqrtf(t) on [0, 46]

  const uint32_t v = ub + 0x20000000u;
  const int32_t g = static_cast<int32_t>(v & 0x3fffffffu) - 0x20000000;
  const float a = static_cast<float>(g) * 0x1.921fb6p-30f;  // alpha in [-pi/4, pi/4)
  const float z = a * a;
  float ps = __builtin_fmaf(-1.9515295891E-4f, z, 8.3321608736E-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611E-1f);
  const float az = a * z;
  const float s = __builtin_fmaf(az, ps, a);
  float pc = __builtin_fmaf(2.443315711809948E-5f, z, -1.388731625493765E-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827E-2f);
  const float zz = z * z;
  const float h = __builtin_fmaf(-0.5f, z, 1.0f);
  const float c = __builtin_fmaf(zz, pc, h);

  const bool swap = (v & 0x40000000u) != 0;
  const uint32_t sign_s = v & 0x80000000u;
  const uint32_t sign_c = (v + 0x40000000u) & 0x80000000u;
  const float cb = swap ? s : c;
  const float sb = swap ? c : s;
  z_cos = r * __uint_as_float(__float_as_uint(cb) ^ sign_c);
  z_sin = r * __uint_as_float(__float_as_uint(sb) ^ sign_s);
}

// Draws per Philox block: 8 for table mode with T <= 2048 ("dense"), else 4.
template <int kMode, bool kDense>
struct Draws {
  static constexpr int value = (kMode == SMMC_MODE_TABLE && kDense) ? 8 : 4;
};

// Four base-T digits of the 64-bit fraction (h:l): digit k = floor(T * frac(T^k x)) by
// exact 64 x 32-bit multiplies (two v_mad_u64_u32 each) for k = 0..2, the last one from
// the top 32 bits of what is left.  Relative bias < T^4/2^64 + T/2^32 (< 1.3e-6, T <= 2048).
__device__ __forceinline__ void digits4(uint32_t h, uint32_t l, uint32_t T, uint32_t (&idx)[4]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const uint64_t pl = static_cast<uint64_t>(l) * T;
    const uint64_t ph = static_cast<uint64_t>(h) * T + (pl >> 32);
    idx[d] = static_cast<uint32_t>(ph >> 32);
    h = static_cast<uint32_t>(ph);
    l = static_cast<uint32_t>(pl);
  }
  idx[3] = __umulhi(h, T);
}

// The per-period multipliers a = 100.0f + r of Philox block `blk` of a path.
template <int kMode, bool kDense>
__device__ __forceinline__ void block_multipliers(const KernelArgs &k, const float *lds_table,
                                                  uint32_t path_lo, uint32_t path_hi, uint32_t blk,
                                                  float (&a)[Draws<kMode, kDense>::value]) {
  uint32_t u[4];
  philox4x32_10(path_lo, path_hi, blk, static_cast<uint32_t>(kMode), k.key0, k.key1, u);
  if constexpr (kMode == SMMC_MODE_TABLE && kDense) {
    uint32_t ia[4], ib[4];
    digits4(u[0], u[1], k.table_len, ia);
    digits4(u[2], u[3], k.table_len, ib);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = lds_table[ia[j]];
      a[4 + j] = lds_table[ib[j]];
    }
  } else if constexpr (kMode == SMMC_MODE_TABLE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = lds_table[__umulhi(u[j], k.table_len)];
  } else {
    float z[4];
    box_muller(u[0], u[1], z[0], z[1]);
    box_muller(u[2], u[3], z[2], z[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = 100.0f + __builtin_fmaf(k.gauss_std, z[j], k.gauss_mean);
  }
}

template <int kMode, bool kExactDiv, bool kDense>
__device__ __forceinline__ float simulate_path(const KernelArgs &k, const float *lds_table,
                                               uint64_t path) {
  constexpr int kDraws = Draws<kMode, kDense>::value;
  const uint32_t path_lo = static_cast<uint32_t>(path);
  const uint32_t path_hi = static_cast<uint32_t>(path >> 32);
  float total = k.initial_capital;
  const uint32_t full = k.n_periods / kDraws;
  for (uint32_t blk = 0; blk < full; ++blk) {
    float a[kDraws];
    block_multipliers<kMode, kDense>(k, lds_table, path_lo, path_hi, blk, a);
#pragma unroll
    for (int j = 0; j < kDraws; ++j) total = compound<kExactDiv>(total, a[j]);
  }
  const uint32_t rem = k.n_periods - full * kDraws;
  if (rem) {  // wave-uniform
    float a[kDraws];
    block_multipliers<kMode, kDense>(k, lds_table, path_lo, path_hi, full, a);
#pragma unroll
    for (int j = 0; j < kDraws - 1; ++j)
      if (static_cast<uint32_t>(j) < rem) total = compound<kExactDiv>(total, a[j]);
  }
  return total;
}

// ---- reductions: wave shuffles, then LDS across the 4 waves ------------------

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

constexpr int kWaves = kBlock / 64;

// Sum over the workgroup, result broadcast to every thread.  Fixed order:
// shuffle tree inside each wave, then wave 0 + 1 + 2 + 3.
__device__ __forceinline__ double block_sum_bcast(double v, double *scratch /* kWaves + 1 */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = wave_sum(v);
  __syncthreads();  // scratch may still be read from the previous use
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = scratch[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) t += scratch[w];
    scratch[kWaves] = t;
  }
  __syncthreads();
  return scratch[kWaves];
}

// ---- main kernel ---------------------------------------------------------------

// Persistent workgroups; each iteration one chunk of 256 consecutive paths (one
// per lane) so the final-value store of a wave is one 256-byte line-aligned
// segment.  LDS: [table (100 + r)] [histogram u32 bins].
template <int kMode, bool kExactDiv, bool kDense>
__global__ __launch_bounds__(kBlock) void paths_kernel(const KernelArgs k) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  float *lds_table = reinterpret_cast<float *>(lds_raw);
  const uint32_t table_words = (kMode == SMMC_MODE_TABLE) ? k.table_len : 0u;
  uint32_t *lds_hist = reinterpret_cast<uint32_t *>(lds_raw) + table_words;
  __shared__ double red_scratch[kWaves + 1];
  __shared__ BlockPartial wave_part[kWaves];

  const uint32_t tid = threadIdx.x;
  if constexpr (kMode == SMMC_MODE_TABLE) {
    for (uint32_t i = tid; i < k.table_len; i += kBlock) lds_table[i] = k.table_a[i];
  }
  const bool want_stats = k.partials != nullptr;
  const bool want_hist = want_stats && k.n_bins != 0;
  if (want_hist) {
    for (uint32_t i = tid; i < k.n_bins; i += kBlock) lds_hist[i] = 0u;
  }
  __syncthreads();

  double sum = 0.0, sumsq = 0.0;
  uint32_t n_count = 0, n_below = 0, n_under = 0, n_over = 0;
  float vmin = __builtin_inff(), vmax = -__builtin_inff();

  const uint64_t n_chunks = (k.n_paths + kBlock - 1) / kBlock;
  for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint64_t i = chunk * kBlock + tid;
    const bool active = i < k.n_paths;
    float v = 0.0f;
    if (active) {
      v = simulate_path<kMode, kExactDiv, kDense>(k, lds_table, k.first_path + i);
      if (k.d_final) k.d_final[i] = v;
    }
    if (want_stats && active) {
      const double dv = static_cast<double>(v);
      sum += dv;
      sumsq += dv * dv;
      n_count += 1;
      n_below += (v < k.below_threshold) ? 1u : 0u;
      vmin = fminf(vmin, v);
      vmax = fmaxf(vmax, v);
      if (want_hist) {
        if (v < k.hist_lo) {
          n_under += 1;
        } else if (v < k.hist_hi) {
          int32_t b = static_cast<int32_t>((dv - static_cast<double>(k.hist_lo)) * k.hist_inv);
          b = b < static_cast<int32_t>(k.n_bins) - 1 ? b : static_cast<int32_t>(k.n_bins) - 1;
          atomicAdd(&lds_hist[b], 1u);
        } else {
          n_over += 1;
        }
      }
    }
    if (k.d_chunk_mean || k.d_chunk_var) {  // wave-uniform
      const uint64_t left = k.n_paths - chunk * kBlock;
      const double n_in = static_cast<double>(left < kBlock ? left : kBlock);
      const double dv = active ? static_cast<double>(v) : 0.0;
      const double mean = block_sum_bcast(dv, red_scratch) / n_in;
      const double d = active ? dv - mean : 0.0;
      const double var = block_sum_bcast(d * d, red_scratch) / n_in;
      if (tid == 0) {
        if (k.d_chunk_mean) k.d_chunk_mean[chunk] = static_cast<float>(mean);
        if (k.d_chunk_var) k.d_chunk_var[chunk] = static_cast<float>(var);
      }
    }
  }

  if (want_stats) {
    // per-lane u32 counters cannot overflow: a lane sees < 2^32 chunks
    BlockPartial p;
    p.sum = wave_sum(sum);
    p.sumsq = wave_sum(sumsq);
    p.count = wave_sum(static_cast<unsigned long long>(n_count));
    p.below = wave_sum(static_cast<unsigned long long>(n_below));
    p.underflow = wave_sum(static_cast<unsigned long long>(n_under));
    p.overflow = wave_sum(static_cast<unsigned long long>(n_over));
    p.min = wave_min(vmin);
    p.max = wave_max(vmax);
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) wave_part[wave] = p;
    __syncthreads();
    if (tid == 0) {
      BlockPartial t = wave_part[0];
#pragma unroll
      for (int w = 1; w < kWaves; ++w) {
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
    if (want_hist) {  // LDS atomics of all waves are complete after the barrier above
      for (uint32_t b = tid; b < k.n_bins; b += kBlock) {
        const uint32_t c = lds_hist[b];
        if (c) atomicAdd(&k.d_hist[b], static_cast<unsigned long long>(c));
      }
    }
  }
}

// Reduces the per-workgroup partials in a fixed order and writes the record header.
__global__ __launch_bounds__(kBlock) void finalize_kernel(const BlockPartial *partials,
                                                          uint32_t n_partials, smmc_stats *out,
                                                          uint32_t n_bins) {
  __shared__ BlockPartial sh[kBlock];
  BlockPartial t;
  t.sum = 0.0;
  t.sumsq = 0.0;
  t.count = t.below = t.underflow = t.overflow = 0ull;
  t.min = __builtin_inff();
  t.max = -__builtin_inff();
  for (uint32_t j = threadIdx.x; j < n_partials; j += kBlock) {
    const BlockPartial p = partials[j];
    t.sum += p.sum;
    t.sumsq += p.sumsq;
    t.count += p.count;
    t.below += p.below;
    t.underflow += p.underflow;
    t.overflow += p.overflow;
    t.min = fminf(t.min, p.min);
    t.max = fmaxf(t.max, p.max);
  }
  sh[threadIdx.x] = t;
  __syncthreads();
  for (uint32_t s = kBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      BlockPartial &a = sh[threadIdx.x];
      const BlockPartial &b = sh[threadIdx.x + s];
      a.sum += b.sum;
      a.sumsq += b.sumsq;
      a.count += b.count;
      a.below += b.below;
      a.underflow += b.underflow;
      a.overflow += b.overflow;
      a.min = fminf(a.min, b.min);
      a.max = fmaxf(a.max, b.max);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const BlockPartial &r = sh[0];
    out->count = r.count;
    out->below = r.below;
    out->underflow = r.underflow;
    out->overflow = r.overflow;
    out->sum = r.sum;
    out->sumsq = r.sumsq;
    out->min = r.min;
    out->max = r.max;
    out->n_bins = n_bins;
    out->reserved = 0;
  }
}

// ---- keepdata: every trajectory, path-major ------------------------------------
//
// mc_data of mc_simulations_keepdata (src/simulations.cpp:139-186): row i holds the
// n_periods + 1 values of path i (values[0] = initial capital).  This one IS
// HBM-bound: 4 (P + 1) bytes per path.
//
// A lane owns a path, so one period's values of a wave form a COLUMN of the output;
// rows are 4 (P + 1) bytes apart, i.e. start at arbitrary 4-byte offsets.  Writing
// tile-aligned runs leaves partial 128-byte lines in L2 for a whole tile time, and
// their in-flight footprint (~ waves x 16 KB) is the size of the aggregate L2: that
// version ran at 2.7 TB/s.  Here every lane runs its path with its own DELAY of phi
// columns, phi = (float address of its row start) mod kTile, so that column c of
// tile t of EVERY row lands on a float address that is c mod kTile: each wave-wide
// store covers whole aligned kTile*4-byte windows (256 or 128 bytes), complete lines
// the moment they are written.  Only the first and last window of a row are partial.
//
// Column g of lane r holds value index s = g - phi_r (0 = initial capital, s >= 1 =
// after period s - 1).  Philox blocks stay wave-uniform in time: with D draws per
// block, at column group G every lane draws block G - q_r, and a log2(D)-stage barrel
// shifter over the previous and current block's multipliers applies the lane's
// residual shift rho_r in [0, D) (phi_r + 1 = D q_r + rho_r).  Lanes idle (keep their
// value) outside 1 <= s <= P.
template <int kMode, bool kExactDiv, bool kDense, int kTile>
__global__ __launch_bounds__(kBlock) void keepdata_kernel(const KernelArgs k) {
  constexpr int kDraws = Draws<kMode, kDense>::value;
  constexpr int kTilePad = kTile + 1;        // +1 word: column writes and row reads conflict-free
  constexpr int kRowsPerStore = 64 / kTile;  // rows covered by one wave-wide store
  extern __shared__ __align__(16) unsigned char lds_raw[];
  float *lds_table = reinterpret_cast<float *>(lds_raw);
  const uint32_t table_words = (kMode == SMMC_MODE_TABLE) ? k.table_len : 0u;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float *tile = lds_table + table_words + wave * (64 * kTilePad);
  if constexpr (kMode == SMMC_MODE_TABLE) {
    for (uint32_t i = tid; i < k.table_len; i += kBlock) lds_table[i] = k.table_a[i];
  }
  __syncthreads();

  const uint32_t sub_row = lane / kTile, col = lane % kTile;  // this lane's place in a store
  const uint32_t n_periods = k.n_periods;
  const uint64_t row_len = static_cast<uint64_t>(n_periods) + 1;
  const uint64_t base_f = reinterpret_cast<uintptr_t>(k.d_traj) >> 2;  // float address of d_traj[0]
  const uint32_t n_tiles = static_cast<uint32_t>((row_len + (kTile - 1) + (kTile - 1)) / kTile);
  const uint64_t n_chunks = (k.n_paths + kBlock - 1) / kBlock;
  for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint64_t wave_first = chunk * kBlock + wave * 64;  // first path of this wave
    if (wave_first >= k.n_paths) continue;                   // wave-uniform
    const uint64_t left = k.n_paths - wave_first;
    const uint32_t rows_here = left < 64 ? static_cast<uint32_t>(left) : 64u;
    const uint64_t i = wave_first + lane;
    const bool active = lane < rows_here;
    // lanes past the end still run a (discarded) path so the wave stays converged
    const uint64_t path = k.first_path + i;
    const uint32_t path_lo = static_cast<uint32_t>(path), path_hi = static_cast<uint32_t>(path >> 32);

    const uint32_t phi = static_cast<uint32_t>(base_f + i * row_len) & (kTile - 1);
    const uint32_t psi = phi + 1;           // period = column - psi
    const uint32_t q = psi / kDraws;        // whole blocks of delay
    const uint32_t rho = psi % kDraws;      // residual shift inside a block

    float total = k.initial_capital;
    float w[2 * kDraws];  // previous block's multipliers, then the current block's
#pragma unroll
    for (int j = 0; j < 2 * kDraws; ++j) w[j] = 100.0f;
    for (uint32_t t = 0; t < n_tiles; ++t) {
      const uint32_t g0 = t * kTile;
      for (uint32_t c = 0; c < kTile; c += kDraws) {
        float cur[kDraws];
        block_multipliers<kMode, kDense>(k, lds_table, path_lo, path_hi, (g0 + c) / kDraws - q, cur);
#pragma unroll
        for (int j = 0; j < kDraws; ++j) {
          w[j] = w[kDraws + j];
          w[kDraws + j] = cur[j];
        }
        // barrel shifter: m[j] = w[kDraws - rho + j], one stage per bit of rho
        float x[2 * kDraws];
#pragma unroll
        for (int j = 0; j < 2 * kDraws; ++j) x[j] = w[j];
#pragma unroll
        for (int sh = 1; sh < kDraws; sh <<= 1) {
          const bool on = (rho & sh) != 0;
#pragma unroll
          for (int j = 2 * kDraws - 1; j >= sh; --j) x[j] = on ? x[j - sh] : x[j];
        }
#pragma unroll
        for (int j = 0; j < kDraws; ++j) {
          const uint32_t period = g0 + c + j - psi;  // wraps below zero -> fails the test
          const float next = compound<kExactDiv>(total, x[kDraws + j]);
          total = period < n_periods ? next : total;
          tile[lane * kTilePad + c + j] = total;
        }
      }
      // The tile is private to this wave: order LDS writes before the row reads.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      {
        const float *src = tile + sub_row * kTilePad + col;
        uint64_t row_off = (wave_first + sub_row) * row_len;  // float offset of this lane's row in d_traj
        const uint64_t row_step = static_cast<uint64_t>(kRowsPerStore) * row_len;
        constexpr int kUnroll = 4;
        uint32_t r = 0;
        for (; r + kUnroll * kRowsPerStore <= rows_here; r += kUnroll * kRowsPerStore) {
          float v[kUnroll];
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) v[u] = src[(r + u * kRowsPerStore) * kTilePad];
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            const uint32_t row_phi = static_cast<uint32_t>(base_f + row_off) & (kTile - 1);
            const uint32_t s_idx = g0 + col - row_phi;  // value index; wraps below zero
            if (s_idx <= n_periods) k.d_traj[row_off + s_idx] = v[u];
            row_off += row_step;
          }
        }
        for (; r < rows_here; r += kRowsPerStore) {
          const uint32_t row_phi = static_cast<uint32_t>(base_f + row_off) & (kTile - 1);
          const uint32_t s_idx = g0 + col - row_phi;
          if (r + sub_row < rows_here && s_idx <= n_periods) k.d_traj[row_off + s_idx] = src[r * kTilePad];
          row_off += row_step;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
    if (active && k.d_final) k.d_final[i] = total;
  }
}

// Device self-test of the two arithmetic shortcuts against the compiler's IEEE forms,
// over the binary32 patterns [lo, hi): counts[0] += mismatches of div100<false> vs
// x / 100.0f, counts[1] += mismatches of sqrt_rn_pos vs sqrtf.
__global__ __launch_bounds__(kBlock) void selftest_kernel(uint32_t lo, uint32_t hi,
                                                          unsigned long long *counts) {
  unsigned long long bad_div = 0, bad_sqrt = 0;
  const uint64_t n = static_cast<uint64_t>(hi) - lo;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * kBlock) {
    const float x = __uint_as_float(lo + static_cast<uint32_t>(i));
    if (__float_as_uint(div100<false>(x)) != __float_as_uint(x / 100.0f)) ++bad_div;
    if (__float_as_uint(sqrt_rn_pos(x)) != __float_as_uint(__builtin_sqrtf(x))) ++bad_sqrt;
  }
  bad_div = wave_sum(bad_div);
  bad_sqrt = wave_sum(bad_sqrt);
  if ((threadIdx.x & 63) == 0) {
    if (bad_div) atomicAdd(&counts[0], bad_div);
    if (bad_sqrt) atomicAdd(&counts[1], bad_sqrt);
  }
}

}  // namespace

hipError_t launch_selftest(uint32_t lo, uint32_t hi, unsigned long long *d_counts, uint32_t grid,
                           hipStream_t stream) {
  hipLaunchKernelGGL(selftest_kernel, dim3(grid), dim3(kBlock), 0, stream, lo, hi, d_counts);
  return hipGetLastError();
}

size_t paths_lds_bytes(uint32_t table_len, uint32_t n_bins) {
  return (static_cast<size_t>(table_len) + n_bins) * 4u;
}
size_t keepdata_lds_bytes(uint32_t table_len, int tile) {
  return (static_cast<size_t>(table_len) + static_cast<size_t>(kWaves) * 64 * (tile + 1)) * 4u;
}

template <int kMode, bool kDense>
static hipError_t launch_paths_mode(const KernelArgs &a, bool exact_div, uint32_t grid, size_t lds,
                                    hipStream_t stream) {
  if (exact_div)
    hipLaunchKernelGGL((paths_kernel<kMode, true, kDense>), dim3(grid), dim3(kBlock), lds, stream, a);
  else
    hipLaunchKernelGGL((paths_kernel<kMode, false, kDense>), dim3(grid), dim3(kBlock), lds, stream, a);
  return hipGetLastError();
}

bool table_is_dense(uint32_t table_len) { return table_len <= kDenseMaxTable; }

hipError_t launch_paths(const KernelArgs &a, bool exact_div, uint32_t grid, size_t lds_bytes,
                        hipStream_t stream) {
  if (a.mode != SMMC_MODE_TABLE) return launch_paths_mode<SMMC_MODE_GAUSSIAN, false>(a, exact_div, grid, lds_bytes, stream);
  return table_is_dense(a.table_len) ? launch_paths_mode<SMMC_MODE_TABLE, true>(a, exact_div, grid, lds_bytes, stream)
                                     : launch_paths_mode<SMMC_MODE_TABLE, false>(a, exact_div, grid, lds_bytes, stream);
}

hipError_t launch_finalize(const BlockPartial *partials, uint32_t n_partials, smmc_stats *d_stats,
                           uint32_t n_bins, hipStream_t stream) {
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kBlock), 0, stream, partials, n_partials, d_stats,
                     n_bins);
  return hipGetLastError();
}

template <int kMode, bool kDense, int kTile>
static hipError_t launch_keepdata_tile(const KernelArgs &a, bool exact_div, uint32_t grid, size_t lds,
                                       hipStream_t stream) {
  if (exact_div)
    hipLaunchKernelGGL((keepdata_kernel<kMode, true, kDense, kTile>), dim3(grid), dim3(kBlock), lds, stream, a);
  else
    hipLaunchKernelGGL((keepdata_kernel<kMode, false, kDense, kTile>), dim3(grid), dim3(kBlock), lds, stream, a);
  return hipGetLastError();
}

template <int kMode, bool kDense>
static hipError_t launch_keepdata_mode(const KernelArgs &a, bool exact_div, int tile, uint32_t grid,
                                       hipStream_t stream) {
  const size_t lds = keepdata_lds_bytes(a.table_len, tile);
  switch (tile) {
    case 32: return launch_keepdata_tile<kMode, kDense, 32>(a, exact_div, grid, lds, stream);
    case 64: return launch_keepdata_tile<kMode, kDense, 64>(a, exact_div, grid, lds, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_keepdata(const KernelArgs &a, bool exact_div, int tile, uint32_t grid, hipStream_t stream) {
  if (a.mode != SMMC_MODE_TABLE) return launch_keepdata_mode<SMMC_MODE_GAUSSIAN, false>(a, exact_div, tile, grid, stream);
  return table_is_dense(a.table_len) ? launch_keepdata_mode<SMMC_MODE_TABLE, true>(a, exact_div, tile, grid, stream)
                                     : launch_keepdata_mode<SMMC_MODE_TABLE, false>(a, exact_div, tile, grid, stream);
}

}  // namespace smmc
