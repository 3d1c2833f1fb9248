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

#include "smmc_device.h"
#include "smmc_internal.h"

namespace smmc {
namespace {

using namespace dev;

constexpr uint32_t kPhiloxM0 = 0xD2511F53u;
constexpr uint32_t kPhiloxM1 = 0xCD9E8D57u;
constexpr uint32_t kWeyl0 = 0x9E3779B9u;
constexpr uint32_t kWeyl1 = 0xBB67AE85u;
constexpr uint32_t kDenseMaxTable = 2048u;  // largest table drawn eight-per-block

// Philox4x32-10 (Salmon et al., SC'11).  The key schedule is wave-uniform and
// lives in SGPRs; per round the lanes pay two 32x32->64 multiplies
// (v_mad_u64_u32) and two three-input XORs (v_bitop3_b32).
// kUniformFirst: the caller's c0 (counter stream v3: the block index) is wave-uniform and c1, c2 (the
// path) do not change along the caller's loop.  Rounds 0 and 1 then cost two VALU instructions:
//   round 0: M0 c0 is a scalar product, M1 c2 per-path constant; n0 = hi(M1 c2) ^ c1 ^ k0 is
//            per-path constant (hoisted out of the period loop), n2 = hi(M0 c0) ^ c3 ^ k1 scalar;
//   round 1: M0 n0 per-path constant (hoisted), M1 n2 a scalar product; each XOR has ONE per-lane
//            term and a scalar pair, which is formed on the SALU (opaque to the compiler, which
//            otherwise re-associates it apart: an instruction reads at most one SGPR, so a v_bitop3
//            with two scalar terms costs a v_mov); round 2's first XOR likewise.
// With the counter (path, block, mode) of stream v2 the same rounds cost four (one multiply among them).
// Issue rates on gfx950 (tools/ubench_ops.hip, profiles/r04/ubench_ops.jsonl): a VALU instruction that reads an
// SGPR operand issues at HALF rate (v_xor_b32 2.1 clk per wave with VGPR or literal operands, 4.1 with one
// SGPR; v_mul_f32, v_add_u32 the same), and v_bitop3_b32 takes 2.7 clk with three VGPRs against 4.1 with a
// scalar round key.  The wave-uniform constants the period loops read in every block -- the Philox round keys
// of the three-input XORs, the Gaussian draw's additive term -- are therefore held in VGPRs (made opaque to
// the compiler, which would otherwise put every uniform value into an SGPR): 16 + 1 registers per lane, set
// up once per kernel.  Values that change from block to block (the scalarised first rounds) stay scalar: a
// v_mov into a VGPR costs what the SGPR operand costs.
struct DrawRegs {
  uint32_t k0[10], k1[10];  // Philox round keys key + r * Weyl, as VGPRs
  float shift100;           // 100.0f + gauss_mean
};
__device__ __forceinline__ DrawRegs make_draw_regs(const KernelArgs &k) {
  DrawRegs d;
  uint32_t a = k.key0, b = k.key1;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    d.k0[r] = a;
    d.k1[r] = b;
    asm("" : "+v"(d.k0[r]));  // not volatile: set up once, outside the loops; unused ones disappear
    asm("" : "+v"(d.k1[r]));
    a += kWeyl0;
    b += kWeyl1;
  }
  d.shift100 = k.gauss_shift100;
  asm("" : "+v"(d.shift100));
  return d;
}

template <bool kUniformFirst = false>
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, const DrawRegs &dr, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = static_cast<uint64_t>(kPhiloxM0) * c0;
    const uint64_t p1 = static_cast<uint64_t>(kPhiloxM1) * c2;
    const uint32_t h0 = static_cast<uint32_t>(p0 >> 32), h1 = static_cast<uint32_t>(p1 >> 32);
    uint32_t n0, n2;
    if (r == 1 && kUniformFirst) {
      uint32_t pair0 = h1 ^ k0, pair2 = c3 ^ k1;  // scalar ^ scalar
      asm("" : "+s"(pair0));
      asm("" : "+s"(pair2));
      n0 = pair0 ^ c1;
      n2 = pair2 ^ h0;
    } else if (r == 2 && kUniformFirst) {  // c1 = lo(M1 n2 of round 0) is still scalar
      uint32_t pair0 = c1 ^ k0;
      asm("" : "+s"(pair0));
      n0 = pair0 ^ h1;
      n2 = xor3(h0, c3, dr.k1[r]);
    } else if (r < 2) {  // plain XORs: invariant and scalar parts are the compiler's to hoist and to scalarise
      n0 = (h1 ^ k0) ^ c1;
      n2 = (h0 ^ k1) ^ c3;
    } else {
      n0 = xor3(h1, c1, dr.k0[r]);
      n2 = xor3(h0, c3, dr.k1[r]);
    }
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

// N independent Philox4x32-10 blocks, round by round: the source order the compiler keeps is then
// N interleaved dependency chains (left to schedule two calls of philox4x32_10 it runs one after the
// other), so that one wave has 2 N multiplies and 2 N XORs in flight per round instead of 2 + 2.
// A wave alone can issue a dependent VALU instruction only every 4-8 cycles; below ~6 waves per SIMD
// this instruction-level parallelism is what fills the issue slots.
template <int N, bool kUniformFirst = false>
__device__ __forceinline__ void philox4x32_10_multi(uint32_t (&c)[N][4], uint32_t k0, uint32_t k1, const DrawRegs &dr) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0[N], p1[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      p0[i] = static_cast<uint64_t>(kPhiloxM0) * c[i][0];
      p1[i] = static_cast<uint64_t>(kPhiloxM1) * c[i][2];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t h0 = static_cast<uint32_t>(p0[i] >> 32), h1 = static_cast<uint32_t>(p1[i] >> 32);
      uint32_t n0, n2;  // rounds 0 .. 2: see philox4x32_10
      if (r == 1 && kUniformFirst) {
        uint32_t pair0 = h1 ^ k0, pair2 = c[i][3] ^ k1;
        asm("" : "+s"(pair0));
        asm("" : "+s"(pair2));
        n0 = pair0 ^ c[i][1];
        n2 = pair2 ^ h0;
      } else if (r == 2 && kUniformFirst) {
        uint32_t pair0 = c[i][1] ^ k0;
        asm("" : "+s"(pair0));
        n0 = pair0 ^ h1;
        n2 = xor3(h0, c[i][3], dr.k1[r]);
      } else if (r < 2) {
        n0 = (h1 ^ k0) ^ c[i][1];
        n2 = (h0 ^ k1) ^ c[i][3];
      } else {
        n0 = xor3(h1, c[i][1], dr.k0[r]);
        n2 = xor3(h0, c[i][3], dr.k1[r]);
      }
      c[i][1] = static_cast<uint32_t>(p1[i]);
      c[i][3] = static_cast<uint32_t>(p0[i]);
      c[i][0] = n0;
      c[i][2] = n2;
    }
    k0 += kWeyl0;
    k1 += kWeyl1;
  }
}

// Box-Muller of counter stream v2 through two small LDS tables (generated by
// tools/gen_bm_tables.py, staged by the Gaussian kernels at LDS offset 0):
//   [0, 16.5 KiB)       radius: 2 sides x 33 octaves x 16 sub-intervals x {c0, c1, c2, c3}
//   [16.5, 18.5 KiB)    trig:   256 x {cos, sin}(2 pi i / 256)
// r = sqrt(-2 ln U) needs no log and no sqrt: the distance w of the uniform from the end of
// (0,1) it is nearer to is converted to binary32 (v_cvt_f32_u32 of 2w+1); the float's exponent is
// the octave, its top four mantissa bits the sub-interval, the other 19 the cubic's argument.
// The angle 2 pi ub / 2^32 is the table angle of its top 8 bits (rounded) plus |delta| <= pi/256,
// rotated in with sin(delta) = delta (1 - delta^2/6), cos(delta) = 1 - delta^2/2.  Integer steps
// and single IEEE operations only: bit-identical to the oracle's C version.
constexpr uint32_t kBmRadiusEntries = 1056, kBmTrigEntries = 256;
constexpr uint32_t kBmFloats = kBmRadiusEntries * 4 + kBmTrigEntries * 2;
// counter stream v3 (the default Gaussian draw): 2 x 256 radius bins (8 KiB), 2048 sectors (16 KiB)
// (kBm3SubBits, kBm3TrigBits, kBm3AngleK, kBm3AngleC: smmc_internal.h, checked against the generated tables)
constexpr uint32_t kBm3RadiusEntries = 64u << kBm3SubBits, kBm3TrigEntries = 1u << kBm3TrigBits;
constexpr uint32_t kBm3Floats = kBm3RadiusEntries * 4 + kBm3TrigEntries * 2;  // as they lie in global memory
// ... and in LDS (stage_tables): the radius bin's byte offset is (bits >> 16) & kBm3RadiusMask of the
// binary32 pattern of the SIGNED distance -- exponent's low five bits and three mantissa bits at
// [4, 12), the sign (the side) left where the shift puts it, bit 15 -- so side 0 lies at [0, 4 KiB) and
// side 1 at [32, 36 KiB); the angle table fills [4, 20 KiB) between them, [20, 32 KiB) stays unused.
// 36 KiB per workgroup instead of 24 are four workgroups of paths_kernel per CU instead of six: the same
// speed (measured with a padded allocation, profiles/r02/ab_lds_pad.txt; two per CU lose 17 %).
constexpr uint32_t kBm3RadiusMask = 0x8000u | ((32u << kBm3SubBits) - 1u) << 4;
constexpr uint32_t kBm3Side1Bytes = 0x8000u, kBm3SideBytes = (32u << kBm3SubBits) * 16u;
constexpr uint32_t kBm3TrigBytes = kBm3SideBytes;                     // LDS byte address of the angle table
constexpr uint32_t kBm3LdsWords = (kBm3Side1Bytes + kBm3SideBytes) / 4u;
static_assert(kBm3TrigBytes + kBm3TrigEntries * 8u <= kBm3Side1Bytes, "angle table must fit between the radius sides");
// The kernels' kMode template argument: SMMC_MODE_TABLE (0) and SMMC_MODE_GAUSSIAN (1) are counter
// stream v3; kModeGaussianV2 and kModeTableV2 are stream v2 (SMMC_FLAG_STREAM_V2: its counter layout
// and, in Gaussian mode, its draw).  The Philox counter's mode word is 0 for table draws and 1 for
// Gaussian draws in both streams.
constexpr int kModeGaussianV2 = 2, kModeTableV2 = 3;
constexpr bool is_table(int mode) { return mode == SMMC_MODE_TABLE || mode == kModeTableV2; }
// Counter stream v3 (SMMC_MODE_TABLE, SMMC_MODE_GAUSSIAN) counts Philox blocks in the counter's FIRST
// word, (block, path_lo, path_hi, mode); stream v2 (the ...V2 modes) in its third, (path_lo, path_hi,
// block, mode).  With the block first, a wave-uniform block index and a path that does not change
// along the period loop the first two Philox rounds cost two VALU instructions instead of four
// (philox4x32_10).
constexpr bool counter_v3(int mode) { return mode == SMMC_MODE_TABLE || mode == SMMC_MODE_GAUSSIAN; }
constexpr uint32_t mode_tag(int mode) { return is_table(mode) ? 0u : 1u; }
constexpr uint32_t bm_floats(int mode) { return mode == kModeGaussianV2 ? kBmFloats : kBm3Floats; }
// LDS words the mode's Box-Muller tables take (what follows them starts there)
constexpr uint32_t bm_lds_words(int mode) { return mode == kModeGaussianV2 ? kBmFloats : kBm3LdsWords; }

// The two draws fma(r * scale, cos(theta), shift), fma(r * scale, sin(theta), shift), in two steps so
// that several of them can be in flight at once: bm_issue does the integer work and starts the two LDS
// gathers, bm_finish evaluates the cubic and rotates.
struct BmPending {
  float4 kr;      // the radius cubic of the uniform's bin
  float2 cs;      // (cos, sin) of the table angle
  float x;        // the cubic's argument
  float sd, cd;   // sin and cos of the residual angle
};

__device__ __forceinline__ BmPending bm_issue(const float *lds_bm, uint32_t ua, uint32_t ub) {
  BmPending p;
  const uint32_t mask = static_cast<uint32_t>(static_cast<int32_t>(ua) >> 31);  // all ones when U >= 1/2
  const uint32_t w1 = ((ua ^ mask) << 1) | 1u;                                   // 2 w + 1: odd
  const uint32_t bits = __float_as_uint(static_cast<float>(w1));                 // exponent 127 .. 159
  const uint32_t entry = (mask & 528u) + (bits >> 19) - (127u << 4);
  p.x = __uint_as_float(0x3f800000u | ((bits << 4) & 0x007ffff0u)) - 1.5f;
  p.kr = reinterpret_cast<const float4 *>(lds_bm)[entry];

  const uint32_t i = (ub + 0x00800000u) >> 24;
  const int32_t d = static_cast<int32_t>(ub << 8) >> 8;  // low 24 bits, sign-extended
  const float delta = static_cast<float>(d) * 0x1.921fb6p-30f;
  const float d2 = delta * delta;
  p.sd = delta * __builtin_fmaf(d2, -0x1.555556p-3f, 1.0f);
  p.cd = __builtin_fmaf(d2, -0.5f, 1.0f);
  p.cs = reinterpret_cast<const float2 *>(lds_bm + kBmRadiusEntries * 4)[i];
  return p;
}

__device__ __forceinline__ void bm_finish(const BmPending &p, float scale, float shift, float &d_cos, float &d_sin) {
  const float r = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(p.kr.w, p.x, p.kr.z), p.x, p.kr.y), p.x, p.kr.x);
  const float rs = r * scale;
  const float ct = __builtin_fmaf(-p.cs.y, p.sd, p.cs.x * p.cd);
  const float st = __builtin_fmaf(p.cs.x, p.sd, p.cs.y * p.cd);
  d_cos = __builtin_fmaf(rs, ct, shift);
  d_sin = __builtin_fmaf(rs, st, shift);
}

__device__ __forceinline__ void box_muller_scaled(const float *lds_bm, uint32_t ua, uint32_t ub, float scale,
                                                  float shift, float &d_cos, float &d_sin) {
  bm_finish(bm_issue(lds_bm, ua, ub), scale, shift, d_cos, d_sin);
}

// Counter stream v3: the same construction with cheaper arithmetic (6 instead
// of 16.5 VALU per draw).  The first word read as int32 IS the signed distance of the uniform from
// the nearer end of (0, 1); one v_cvt_f32_i32 turns it into a float f whose pattern holds everything:
// the radius bin is the top half of the pattern under a mask -- exponent's low five bits and three
// mantissa bits, the octaves stored rotated, the side left where the float has its sign (one SDWA
// v_and reading the upper word) -- and the cubic is evaluated in f itself (the coefficients carry the
// bin's position, the octave's powers of two, the sign and the factor std).  The angle is the low 30
// bits of the second word: its sector times 8 is again one masked read of the upper word, the table
// has 2048 entries at the MIDDLE of their sectors, and the residual angle -- low bits OR-ed into 1.0f,
// one fma, no conversion -- is small enough for a first-order rotation (c - s delta, s + c delta) (the
// table carries the factor that keeps the mean square length 1); and the draw is the MULTIPLIER
// itself, fma(r std, cos theta, 100 + mean).
struct Bm3Pending {
  float4 kr;
  float2 cs;
  float f, delta;
};

__device__ __forceinline__ Bm3Pending bm3_issue(const float *lds_bm, uint32_t ua, uint32_t ub) {
  Bm3Pending p;
  static_assert(23u - kBm3SubBits - 4u == 16u, "the bin's offset is the top half of the float's pattern");
  p.f = static_cast<float>(static_cast<int32_t>(ua));          // |f| <= 2^31; pattern 0 reads bin 0 (u = 2^-33)
  const uint32_t off = (__float_as_uint(p.f) >> 16) & kBm3RadiusMask;
  (void)lds_bm;
  const f32x4_t kr = lds_load_at<f32x4_t>(off);
  p.kr = make_float4(kr.x, kr.y, kr.z, kr.w);

  constexpr uint32_t kRes = kBm3AngleBits - kBm3TrigBits;      // residual bits below the sector
  static_assert(kRes - 3u == 16u, "sector * 8 is the word's upper half under a mask");
  const uint32_t aoff = (ub >> 16) & ((kBm3TrigEntries - 1u) << 3);  // sector * 8, no rounding add
  // residual angle from the sector's middle, (low bits - half a sector) 2 pi / 2^30: the low bits as the
  // mantissa of a float in [1, 1 + 2^-4), then one fma
  const float ya = __uint_as_float((ub & ((1u << kRes) - 1u)) | 0x3f800000u);
  p.delta = __builtin_fmaf(ya, kBm3AngleK, -kBm3AngleC);
  const f32x2_t cs = lds_load_at<f32x2_t>(aoff + kBm3TrigBytes);
  p.cs = make_float2(cs.x, cs.y);
  return p;
}

// The staged radius coefficients carry the factor std (stage_tables): the cubic IS r std.
__device__ __forceinline__ void bm3_finish(const Bm3Pending &p, float shift, float &d_cos, float &d_sin) {
  const float rs = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(p.kr.w, p.f, p.kr.z), p.f, p.kr.y), p.f, p.kr.x);
  const float ct = __builtin_fmaf(-p.cs.y, p.delta, p.cs.x);
  const float st = __builtin_fmaf(p.cs.x, p.delta, p.cs.y);
  d_cos = __builtin_fmaf(rs, ct, shift);
  d_sin = __builtin_fmaf(rs, st, shift);
}

// Copies the mode's read-only tables from global memory (L2-resident) into LDS.
template <int kMode>
__device__ __forceinline__ void stage_tables(const KernelArgs &k, float *lds, uint32_t block = kBlock) {
  if constexpr (is_table(kMode)) {
    for (uint32_t i = threadIdx.x; i < k.table_len; i += block) lds[i] = k.table_a[i];
  } else {
    const float4 *src = reinterpret_cast<const float4 *>(k.bm_tables);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    for (uint32_t i = threadIdx.x; i < bm_floats(kMode) / 4; i += block) {
      float4 v = src[i];
      uint32_t at = i;
      if (kMode == SMMC_MODE_GAUSSIAN) {
        if (i < kBm3RadiusEntries) {
          // counter stream v3: each radius coefficient times std, rounded once -- the cubic then yields
          // r std and the draw needs no multiply of its own; side 1 goes to its own place (kBm3Side1Bytes)
          v.x *= k.gauss_std;
          v.y *= k.gauss_std;
          v.z *= k.gauss_std;
          v.w *= k.gauss_std;
          if (i >= kBm3RadiusEntries / 2) at = kBm3Side1Bytes / 16u + (i - kBm3RadiusEntries / 2);
        } else {
          at = kBm3TrigBytes / 16u + (i - kBm3RadiusEntries);  // two (cos, sin) pairs per float4
        }
      }
      dst[at] = v;
    }
  }
}

// Draws per Philox block: 8 for table mode with T <= 2048 ("dense"), else 4.
template <int kMode, bool kDense>
struct Draws {
  static constexpr int value = (is_table(kMode) && kDense) ? 8 : 4;
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
template <int kMode, bool kDense, bool kUniformBlock = false>
__device__ __forceinline__ void block_multipliers(const KernelArgs &k, const DrawRegs &dr, const float *lds_table,
                                                  uint32_t path_lo, uint32_t path_hi, uint32_t blk,
                                                  float (&a)[Draws<kMode, kDense>::value]) {
  uint32_t u[4];
  if constexpr (counter_v3(kMode))
    philox4x32_10<kUniformBlock>(blk, path_lo, path_hi, mode_tag(kMode), k.key0, k.key1, dr, u);
  else
    philox4x32_10<false>(path_lo, path_hi, blk, mode_tag(kMode), k.key0, k.key1, dr, u);
  if constexpr (is_table(kMode) && kDense) {
    uint32_t ia[4], ib[4];
    digits4(u[0], u[1], k.table_len, ia);
    digits4(u[2], u[3], k.table_len, ib);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = lds_table[ia[j]];
      a[4 + j] = lds_table[ib[j]];
    }
  } else if constexpr (is_table(kMode)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = lds_table[__umulhi(u[j], k.table_len)];
  } else if constexpr (kMode == kModeGaussianV2) {
    float ret[4];  // the period returns in percent: N(gauss_mean, gauss_std)
    box_muller_scaled(lds_table, u[0], u[1], k.gauss_std, k.gauss_mean, ret[0], ret[1]);
    box_muller_scaled(lds_table, u[2], u[3], k.gauss_std, k.gauss_mean, ret[2], ret[3]);
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = 100.0f + ret[j];
  } else {  // counter stream v3: the multipliers themselves, N(100 + gauss_mean, gauss_std)
    const Bm3Pending p0 = bm3_issue(lds_table, u[0], u[1]);
    const Bm3Pending p1 = bm3_issue(lds_table, u[2], u[3]);
    bm3_finish(p0, dr.shift100, a[0], a[1]);
    bm3_finish(p1, dr.shift100, a[2], a[3]);
  }
}

// The multipliers of the N consecutive Philox blocks blk .. blk + N - 1 of a path, drawn TOGETHER:
// interleaved Philox rounds, then all table gathers issued before the first is used.  The values are
// those of N calls of block_multipliers.
template <int kMode, bool kDense, int N, bool kUniformBlock = false>
__device__ __forceinline__ void block_multipliers_multi(const KernelArgs &k, const DrawRegs &dr, const float *lds_table, uint32_t path_lo,
                                                        uint32_t path_hi, uint32_t blk,
                                                        float (&a)[N][Draws<kMode, kDense>::value]) {
  uint32_t u[N][4];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u[i][0] = counter_v3(kMode) ? blk + i : path_lo;
    u[i][1] = counter_v3(kMode) ? path_lo : path_hi;
    u[i][2] = counter_v3(kMode) ? path_hi : blk + i;
    u[i][3] = mode_tag(kMode);
  }
  philox4x32_10_multi<N, kUniformBlock && counter_v3(kMode)>(u, k.key0, k.key1, dr);
  if constexpr (is_table(kMode) && kDense) {
    uint32_t idx[N][8];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      uint32_t ia[4], ib[4];
      digits4(u[i][0], u[i][1], k.table_len, ia);
      digits4(u[i][2], u[i][3], k.table_len, ib);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        idx[i][j] = ia[j];
        idx[i][4 + j] = ib[j];
      }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) a[i][j] = lds_table[idx[i][j]];
  } else if constexpr (is_table(kMode)) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) a[i][j] = lds_table[__umulhi(u[i][j], k.table_len)];
  } else if constexpr (kMode == kModeGaussianV2) {
    BmPending p[N][2];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      p[i][0] = bm_issue(lds_table, u[i][0], u[i][1]);
      p[i][1] = bm_issue(lds_table, u[i][2], u[i][3]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      float ret[4];
      bm_finish(p[i][0], k.gauss_std, k.gauss_mean, ret[0], ret[1]);
      bm_finish(p[i][1], k.gauss_std, k.gauss_mean, ret[2], ret[3]);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[i][j] = 100.0f + ret[j];
    }
  } else {
    Bm3Pending p[N][2];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      p[i][0] = bm3_issue(lds_table, u[i][0], u[i][1]);
      p[i][1] = bm3_issue(lds_table, u[i][2], u[i][3]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
      bm3_finish(p[i][0], dr.shift100, a[i][0], a[i][1]);
      bm3_finish(p[i][1], dr.shift100, a[i][2], a[i][3]);
    }
  }
}

template <int kMode, int kDiv, bool kDense>
__device__ __forceinline__ float simulate_path(const KernelArgs &k, const DrawRegs &dr, const float *lds_table,
                                               uint64_t path) {
  constexpr int kDraws = Draws<kMode, kDense>::value;
  constexpr bool kExactDiv = kDiv == kDivExact;
  const uint32_t path_lo = static_cast<uint32_t>(path);
  const uint32_t path_hi = static_cast<uint32_t>(path >> 32);
  float total = k.initial_capital;
  bool left_window = false;
  const uint32_t full = k.n_periods / kDraws;
  // One Philox block per trip.  (Two blocks per trip with interleaved rounds -- block_multipliers_multi,
  // what the comb keepdata kernel uses at under 4 waves per SIMD -- ran 6 % SLOWER here: at 6 waves per
  // SIMD the VALU is already saturated once its multi-cycle multiplies are counted, 78 % of the plain
  // issue rate at the clock the chip holds under this load; so did issuing the next block's gathers in
  // front of this block's chain.)
  for (uint32_t blk = 0; blk < full; ++blk) {
    float a[kDraws];
    block_multipliers<kMode, kDense, true>(k, dr, lds_table, path_lo, path_hi, blk, a);
#pragma unroll
    for (int j = 0; j < kDraws; ++j) total = compound<kExactDiv>(total, a[j]);
    if constexpr (kDiv == kDivChecked) left_window |= !(total > k.chk_lo && total < k.chk_hi);  // NaN leaves too
  }
  const uint32_t rem = k.n_periods - full * kDraws;
  if (rem) {  // wave-uniform
    float a[kDraws];
    block_multipliers<kMode, kDense, true>(k, dr, lds_table, path_lo, path_hi, full, a);
#pragma unroll
    for (int j = 0; j < kDraws - 1; ++j)
      if (static_cast<uint32_t>(j) < rem) total = compound<kExactDiv>(total, a[j]);
  }
  if constexpr (kDiv == kDivChecked) {
    if (left_window) total = simulate_path<kMode, kDivExact, kDense>(k, dr, lds_table, path);
  }
  return total;
}

// ---- the workgroups' partial statistics ----------------------------------------------------------
//
// paths_kernel leaves one partial per workgroup; finalize_kernel (one 1024-thread workgroup, a launch later) folds them
// in a fixed order.  Round 4 tried to fold them INSIDE the launch (the workgroup that finishes last
// writes the header) in three forms, all bit-identical to the separate launch and all measured on the 1e6-path step
// (profiles/r04/config0_variants*.txt; the separate launch: 169.5 us per step, kernel 159 us): a release fence per
// workgroup + one arrival counter 178-265 us (the fence waits for every final value queued in the XCD's L2; same-address
// atomics take ~12 ns each); device-scope atomic exchanges instead of the fence + 64 + 1 counters, the last workgroup
// folding all 3907 partials 180.6 us (15 dependent rounds of device-scope loads per thread); a two-level fold -- the
// workgroup that completes a group of 64 folds it, the one that completes the groups folds those -- 172.5 us.  The
// floor is the chain itself: publish, count, load, publish, count, load are six device-scope round trips of ~1.7 us
// across the eight XCDs, as long as the launch boundary they replace.  The separate launch stays.
__device__ __forceinline__ void partial_identity(BlockPartial &t) {
  t.sum = 0.0;
  t.sumsq = 0.0;
  t.count = t.below = t.underflow = t.overflow = 0ull;
  t.min = __builtin_inff();
  t.max = -__builtin_inff();
}
__device__ __forceinline__ void partial_add(BlockPartial &a, const BlockPartial &b) {
  a.sum += b.sum;
  a.sumsq += b.sumsq;
  a.count += b.count;
  a.below += b.below;
  a.underflow += b.underflow;
  a.overflow += b.overflow;
  a.min = fminf(a.min, b.min);
  a.max = fmaxf(a.max, b.max);
}

// ---- main kernel ---------------------------------------------------------------

// Persistent workgroups; each iteration one chunk of 256 consecutive paths (one
// per lane) so the final-value store of a wave is one 256-byte line-aligned
// segment.  LDS: [table (100 + r)] [histogram u32 bins].
template <int kMode, int kDiv, bool kDense>
__global__ __launch_bounds__(kBlock) void paths_kernel(const KernelArgs k) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  float *lds_table = reinterpret_cast<float *>(lds_raw);  // returns table, or the Box-Muller tables
  const uint32_t table_words = is_table(kMode) ? k.table_len : bm_lds_words(kMode);
  uint32_t *lds_hist = reinterpret_cast<uint32_t *>(lds_raw) + table_words;
  // The reduction scratch lives BEHIND the tables in the dynamic allocation (paths_lds_bytes), not in
  // static __shared__ arrays: static LDS is placed first, and a draw table that does not start at LDS
  // address 0 costs one address add per gather (two VALU instructions per Philox block in Gaussian mode).
  const uint32_t scratch_words = (table_words + ((k.partials != nullptr) ? k.n_bins : 0u) + 1u) & ~1u;  // 8-byte aligned
  double *red_scratch = reinterpret_cast<double *>(reinterpret_cast<uint32_t *>(lds_raw) + scratch_words);  // [4 * kWaves]
  BlockPartial *wave_part = reinterpret_cast<BlockPartial *>(red_scratch + 4 * kWaves);                     // [kWaves]

  const uint32_t tid = threadIdx.x;
  bool parity = false;
  unsigned long long clk0 = 0, real0 = 0;
  if (k.clock_probe) {  // uniform; timing instrumentation only
    clk0 = __builtin_amdgcn_s_memtime();
    real0 = __builtin_amdgcn_s_memrealtime();
  }
  stage_tables<kMode>(k, lds_table);
  const bool want_stats = k.partials != nullptr;
  const bool want_hist = want_stats && k.n_bins != 0;
  if (want_hist) {
    for (uint32_t i = tid; i < k.n_bins; i += kBlock) lds_hist[i] = 0u;
  }
  __syncthreads();

  double sum = 0.0, sumsq = 0.0;
  uint32_t n_count = 0, n_below = 0, n_under = 0, n_over = 0;
  float vmin = __builtin_inff(), vmax = -__builtin_inff();
  const DrawRegs dr = make_draw_regs(k);

  const uint64_t n_chunks = (k.n_paths + kBlock - 1) / kBlock;
  for (uint64_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
    const uint64_t i = chunk * kBlock + tid;
    const bool active = i < k.n_paths;
    float v = 0.0f;
    if (active) {
      v = simulate_path<kMode, kDiv, kDense>(k, dr, lds_table, k.first_path + i);
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
      // mean and population variance of this chunk from one pass (sum and sum of squares in
      // double: the cancellation in E[x^2] - mean^2 costs ~1e-15 relative here), one barrier:
      // the scratch slots alternate with the iteration's parity
      const uint64_t left = k.n_paths - chunk * kBlock;
      const double n_in = static_cast<double>(left < kBlock ? left : kBlock);
      const double dv = active ? static_cast<double>(v) : 0.0;
      const double s1 = wave_sum(dv), s2 = wave_sum(dv * dv);
      double *slot = red_scratch + (parity ? 2 * kWaves : 0);
      parity = !parity;
      const int lane = tid & 63, wave = tid >> 6;
      if (lane == 0) {
        slot[wave] = s1;
        slot[kWaves + wave] = s2;
      }
      __syncthreads();
      if (tid == 0) {
        double t1 = slot[0], t2 = slot[kWaves];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) {
          t1 += slot[w];
          t2 += slot[kWaves + w];
        }
        const double mean = t1 / n_in;
        const double var = t2 / n_in - mean * mean;
        if (k.d_chunk_mean) k.d_chunk_mean[chunk] = static_cast<float>(mean);
        if (k.d_chunk_var) k.d_chunk_var[chunk] = static_cast<float>(var > 0.0 ? var : 0.0);
      }
    }
  }

  if (k.clock_probe && tid == 0) {
    atomicAdd(&k.clock_probe[0], __builtin_amdgcn_s_memtime() - clk0);
    atomicAdd(&k.clock_probe[1], __builtin_amdgcn_s_memrealtime() - real0);
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
// One 1024-thread workgroup: 16 k partials (64 workgroups per CU) take ~8 us instead of 26.
constexpr int kFinalizeBlock = 1024;
__global__ __launch_bounds__(kFinalizeBlock) void finalize_kernel(const BlockPartial *partials,
                                                                  uint32_t n_partials, smmc_stats *out,
                                                                  uint32_t n_bins, unsigned long long *hist_acc,
                                                                  uint32_t spread) {
  // The bucket counts were accumulated in the engine's own array (one copy by paths_kernel, kHistSpread copies by
  // values_stats): fold them into the record's and LEAVE THE ARRAY ZERO for the next launch -- the record needs no
  // memset before a launch and the accumulator none after it (two launches per step instead of three).
  if (spread) {
    unsigned long long *hist = reinterpret_cast<unsigned long long *>(out + 1);
    for (uint32_t b = threadIdx.x; b < n_bins; b += kFinalizeBlock) {
      unsigned long long c = 0;
      for (uint32_t r = 0; r < spread; ++r) {
        unsigned long long *src = hist_acc + static_cast<size_t>(r) * n_bins + b;
        const unsigned long long v = *src;
        c += v;
        if (v) *src = 0;
      }
      hist[b] = c;
    }
  }
  __shared__ BlockPartial sh[kFinalizeBlock];
  BlockPartial t;
  partial_identity(t);
  for (uint32_t j = threadIdx.x; j < n_partials; j += kFinalizeBlock) partial_add(t, partials[j]);
  sh[threadIdx.x] = t;
  __syncthreads();
  for (uint32_t s = kFinalizeBlock / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) partial_add(sh[threadIdx.x], sh[threadIdx.x + s]);
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
// store covers whole aligned 128-byte windows, complete lines the moment they are
// written.  Only the first and last window of a row are partial.
//
// Column g of lane r holds value index s = g - phi_r (0 = initial capital, s >= 1 =
// after period s - 1).  Philox blocks stay wave-uniform in time: with D draws per
// block, at column group G every lane draws block G - q_r (phi_r + 1 = D q_r + rho_r),
// and draw j of it belongs to tile column D G + rho_r + j.  The residual shift rho_r is
// the position of the row inside its LDS slots (see kSlots below); a first version
// applied it with a log2(D)-stage register shifter over two blocks (4.25 VALU per
// value, 1.49 ms where this one takes 1.15).
//
// Column groups are of two kinds: INTERIOR groups (every lane inside its own row: no
// predicate at all) and the few general ones at the two ends of the row, where a
// lane's write goes to a trash word when the column is not its own.
//
// Row junctions.  Adjacent rows share the 128-byte line in which one ends and the next
// begins; written as two partial windows a whole path apart in time, that line reaches
// HBM as two masked writes (measured: 3.8 TB/s pattern ceiling instead of 6.2).  So the
// window in which a row ends is completed with the HEAD of the next row and stored whole,
// and the next row's lane skips its own (partial) first window.  The head costs no second
// simulation: the next row is the next LANE, which drew exactly those values in its own
// first window (the phases line up by construction: the next row starts where this one
// ends), so after the first tile every lane hands its first window to the lane before it
// (one wave shuffle per column, once per 64 paths) and keeps its neighbour's in registers
// until its own row ends.  Junctions between waves (lane 63 | lane 0) stay unmerged.  Needs
// rows of at least two windows so that a window holds at most one junction (`merge`).
// An earlier form ran the next row's stream a second time next to the lane's own in the
// tail groups: +15 % arithmetic, and too much for Gaussian mode to gain from merging.
//
// Store phase: a lane takes FOUR columns of one row (16-byte stores, 8 rows x 128 bytes
// per instruction); for interior tiles (all 32 columns of all 64 rows valid) the byte
// offset of row r is ((phi_0 + r L) & ~31) - phi_0 floats from the wave's first row.
constexpr int kKeepdataMaxBlock = 768;  // 12 waves: 3 per SIMD, so up to 168 VGPRs (the kernel keeps a 32-value window per lane)
// slots per tile row of keepdata_kernel, and the words reserved per row (the larger of its two strides)
constexpr int keepdata_slots(int tile, int draws) { return tile + draws; }
constexpr int keepdata_row_words(int tile, int draws) {
  const int slots = keepdata_slots(tile, draws);
  const int odd = slots | 1, even = slots + ((6 - slots % 4) % 4);
  return odd > even ? odd : even;
}
template <int kMode, bool kExactDiv, bool kDense, int kTile>
__global__ __launch_bounds__(kKeepdataMaxBlock) void keepdata_kernel(const KernelArgs k) {
  constexpr int kDraws = Draws<kMode, kDense>::value;
  // A row of the tile has kTile + kDraws slots; logical column c of row r sits in slot
  // kDraws - rho_r + c, which makes the slot a lane WRITES (kDraws + column-group base + j)
  // independent of its rho: bank = (stride * lane + const) mod 32.  (With rho in the write address,
  // bank = (7 + L mod 8) * lane for a stride of 39: 8-way conflicts at L = 361, and the LDS was the
  // bottleneck -- SQ_LDS_BANK_CONFLICT 73 % of SQ_LDS_IDX_ACTIVE.)  Slots from kTile up hold the
  // last group's columns that belong to the next tile and move down by kTile after the store
  // phase.  Row stride: odd when the row length is even, = 2 mod 4 when it is odd (writes then
  // 2-way, which ds_write_b32 absorbs): either way the store phase's reads -- 4 rows x 8 quads
  // per lane group, each row offset by its own rho -- are conflict-free.
  constexpr int kSlots = keepdata_slots(kTile, kDraws);
  constexpr int kStrideOdd = kSlots | 1;
  constexpr int kStrideEven = kSlots + ((6 - kSlots % 4) % 4);  // smallest >= kSlots that is 2 mod 4
  constexpr int kRowMax = keepdata_row_words(kTile, kDraws);
  constexpr int kQuads = kTile / 4;                          // 16-byte pieces of a window
  constexpr int kRowsPerStore = 64 / kQuads;                 // rows one wave-wide store covers
  constexpr uint32_t kGroupsPerTile = kTile / kDraws;        // psi <= kTile: q <= kGroupsPerTile
  // dynamic LDS: [tables][one tile per wave][one trash word per thread]; the host picks the
  // number of waves (smmc_engine_simulate_keepdata: 4 in table mode, 12 in Gaussian mode)
  extern __shared__ __align__(16) unsigned char lds_raw[];
  float *lds_table = reinterpret_cast<float *>(lds_raw);
  const uint32_t table_words = is_table(kMode) ? k.table_len : bm_lds_words(kMode);
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  const uint32_t stride = ((k.n_periods + 1) & 1u) ? kStrideEven : kStrideOdd;  // uniform
  float *tile = lds_table + table_words + wave * (64 * kRowMax);
  float *my_row = tile + lane * stride;
  float *my_trash = lds_table + table_words + n_waves * (64 * kRowMax) + tid;
  stage_tables<kMode>(k, lds_table, blockDim.x);
  __syncthreads();
  const DrawRegs dr = make_draw_regs(k);

  const uint32_t sub = lane / kQuads, quad = lane % kQuads;  // store phase: row sub + kRowsPerStore it, columns 4 quad .. + 3
  const uint32_t n_periods = k.n_periods;
  const uint32_t row_len32 = n_periods + 1;  // the host keeps n_periods below 2^24 for keepdata
  const uint64_t row_len = row_len32;
  const uint64_t base_f = reinterpret_cast<uintptr_t>(k.d_traj) >> 2;  // float address of d_traj[0]
  const bool merge = row_len32 >= 2u * kTile;  // uniform
  // groups [kGroupsPerTile, tail_start) are interior: 0 <= period < n_periods for every lane
  const uint32_t tail_start = n_periods / kDraws;
  const bool small_rows = row_len32 <= (1u << 22);  // 64 rows of bytes fit 32-bit offsets
  // a wave takes 64 consecutive paths at a time; no workgroup-level step after the staging barrier
  const uint64_t n_wave_chunks = (k.n_paths + 63) / 64;
  for (uint64_t wc = static_cast<uint64_t>(blockIdx.x) * n_waves + wave; wc < n_wave_chunks;
       wc += static_cast<uint64_t>(gridDim.x) * n_waves) {
    const uint64_t wave_first = wc * 64;  // first path of this wave
    const uint64_t left = k.n_paths - wave_first;
    const uint32_t rows_here = left < 64 ? static_cast<uint32_t>(left) : 64u;
    const uint64_t i = wave_first + lane;
    const bool active = lane < rows_here;
    // lanes past the end still run a (discarded) path so the wave stays converged
    const uint64_t path = k.first_path + i;
    const uint32_t path_lo = static_cast<uint32_t>(path), path_hi = static_cast<uint32_t>(path >> 32);

    const uint32_t phi0 = static_cast<uint32_t>(base_f + wave_first * row_len) & (kTile - 1);  // uniform
    const uint32_t phi = (phi0 + lane * row_len32) & (kTile - 1);
    const uint32_t psi = phi + 1;             // own period = column - psi
    const uint32_t q = psi / kDraws;          // whole blocks of delay
    const uint32_t rho = psi % kDraws;        // residual shift: the row's slot offset is kDraws - rho
    const uint32_t g_next = phi + row_len32;  // column that holds values[0] of the next row
    // columns this wave has to draw: up to the last value of its most delayed row (all 32 phases
    // occur in a wave unless the row length shares a factor with the window: rows of 64 values on
    // an aligned base need two tiles, not three)
    uint32_t phi_max = phi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t other = __shfl_xor(phi_max, off, 64);
      phi_max = other > phi_max ? other : phi_max;
    }
    const uint32_t last_col = phi_max + n_periods;
    const uint32_t n_tiles_w = last_col / kTile + 1u;  // <= n_tiles
    float *wr = my_row + kDraws;                 // own draws: slot kDraws + cb + j
    float *logical = my_row + kDraws - rho;      // logical column 0
    float *wave_base = k.d_traj + wave_first * row_len;

    // The store phase's view of the chunk, fixed over its tiles: iteration `it` of a lane handles row
    // r = sub + it * kRowsPerStore.  st_lds: where the row's four columns sit in the LDS tile;
    // st_off: byte offset from the wave's first row to those columns in the row's window 0 (can be
    // negative by up to 4 phi0 for the first row); st_span: first column the row stores << 26 |
    // number of columns it stores (0 for rows past the end of the launch).  A row goes to memory
    // from its first WHOLE window (the lane of the row before wrote the shared one) through the
    // end of its last window (which includes the next row's head), except across waves and at the
    // two ends of the launch.
    uint32_t st_lds[kQuads], st_off[kQuads], st_span[kQuads];
    {
      const uint32_t rows_left = left < 0xFFFFFFFFull ? static_cast<uint32_t>(left) : 0xFFFFFFFFu;
#pragma unroll
      for (int it = 0; it < kQuads; ++it) {
        const uint32_t r = sub + it * kRowsPerStore;
        const uint32_t tf = phi0 + r * row_len32;
        const uint32_t phi_r = tf & (kTile - 1);
        const uint32_t last = phi_r + n_periods;
        const uint32_t lo = (!merge || phi_r == 0 || r == 0) ? phi_r : static_cast<uint32_t>(kTile);
        const uint32_t hi = (!merge || r == 63 || r + 1 >= rows_left) ? last : (last | (kTile - 1));
        st_lds[it] = r * stride + (kDraws + 4u * quad - ((phi_r + 1u) & (kDraws - 1u)));
        st_off[it] = (tf - phi_r - phi0 + 4u * quad) * 4u;
        st_span[it] = r < rows_here ? (lo << 26) | (hi - lo + 1u) : 0u;
      }
    }
    float total = k.initial_capital;
    float next_head[kTile];  // the next row's first window (merge)
    logical[phi] = k.initial_capital;  // values[0]
    for (uint32_t t = 0; t < n_tiles_w; ++t) {
      const uint32_t g0 = t * kTile;
      const uint32_t grp0 = g0 / kDraws;
      const bool interior_tile = grp0 >= kGroupsPerTile && grp0 + kGroupsPerTile <= tail_start;  // uniform
      float last[kDraws];  // the tile's last group (interior tiles)
      if (interior_tile) {
        // every group of the tile is interior: straight-line code, so that the next group's
        // Philox rounds and table reads overlap this group's compounding chain (the kernel runs
        // at 2-3 waves per SIMD, LDS-limited: instruction-level parallelism has to hide latency)
#pragma unroll
        for (uint32_t cb = 0; cb < kTile; cb += kDraws) {
          float a[kDraws];
          block_multipliers<kMode, kDense>(k, dr, lds_table, path_lo, path_hi, grp0 + cb / kDraws - q, a);
#pragma unroll
          for (int j = 0; j < kDraws; ++j) {
            total = compound<kExactDiv>(total, a[j]);
            wr[cb + j] = total;
            if (cb == kTile - kDraws) last[j] = total;
          }
        }
      } else
      for (uint32_t cb = 0; cb < kTile; cb += kDraws) {
        if (g0 + cb > last_col) break;  // uniform: nothing to draw beyond
        const uint32_t grp = (g0 + cb) / kDraws;
        float a[kDraws];
        block_multipliers<kMode, kDense>(k, dr, lds_table, path_lo, path_hi, grp - q, a);
        if (grp >= kGroupsPerTile && grp < tail_start) {  // uniform: every lane is inside its own row
          float *dst = wr + cb;
#pragma unroll
          for (int j = 0; j < kDraws; ++j) {
            total = compound<kExactDiv>(total, a[j]);
            dst[j] = total;
          }
        } else {
          const uint32_t p0 = kDraws * (grp - q);  // wraps below zero: fails the test
#pragma unroll
          for (int j = 0; j < kDraws; ++j) {
            const bool own = p0 + j < n_periods;
            const float next = compound<kExactDiv>(total, a[j]);
            total = own ? next : total;
            float *dst = own ? wr + cb + j : my_trash;
            *dst = total;
          }
        }
      }
      // The tile is private to this wave: order LDS writes before the row reads.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      if (merge && t == 0) {  // uniform: every lane's first window goes to the lane of the row before
#pragma unroll
        for (uint32_t c = 0; c < kTile; ++c) next_head[c] = __shfl_down(logical[c], 1, 64);
      }
      char *base_b = reinterpret_cast<char *>(wave_base);
      const uint32_t g0_b = g0 * 4u, g_first = g0 + 4u * quad;
      if (rows_here == 64 && small_rows && g0 >= kTile && g0 + (kTile - 1) <= n_periods) {
        // interior tile: every column of every row is that row's own
#pragma unroll
        for (int it = 0; it < kQuads; ++it) {
          const float *s = tile + st_lds[it];
          const float4 v = make_float4(s[0], s[1], s[2], s[3]);
          const uint32_t boff = st_off[it] + g0_b;  // in 32 bits: the true, non-negative offset (g0 > phi0)
          *reinterpret_cast<float4 *>(base_b + boff) = v;
        }
      } else if (small_rows) {
        // a tile in which rows begin or end: column g of the row is stored iff g - lo < count
        // (unsigned: columns before lo wrap to huge values)
#pragma unroll
        for (int it = 0; it < kQuads; ++it) {
          const uint32_t count = st_span[it] & 0x03FFFFFFu, d = g_first - (st_span[it] >> 26);
          const float *s = tile + st_lds[it];
          const uint32_t boff = st_off[it] + g0_b;  // non-negative whenever the element is stored
          if (d < count && d + 3u < count) {
            *reinterpret_cast<float4 *>(base_b + boff) = make_float4(s[0], s[1], s[2], s[3]);
          } else {
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e)
              if (d + e < count) *reinterpret_cast<float *>(base_b + (boff + 4u * e)) = s[e];
          }
        }
      } else {  // rows of more than 2^22 values: 64-bit offsets
        const uint32_t rho_mask = kDraws - 1;
        for (uint32_t r = sub; r < rows_here; r += kRowsPerStore) {
          const uint32_t phi_r = (phi0 + r * row_len32) & (kTile - 1);
          const uint64_t row_global = wave_first + r;
          const uint32_t last = phi_r + n_periods;
          const uint32_t lo = (!merge || phi_r == 0 || r == 0) ? phi_r : static_cast<uint32_t>(kTile);
          const uint32_t hi = (!merge || r == 63 || row_global + 1 >= k.n_paths) ? last : (last | (kTile - 1));
          float *row_ptr = wave_base + (static_cast<int64_t>(r) * row_len32 - phi_r);
          const float *s = tile + r * stride + (kDraws + 4u * quad - ((phi_r + 1u) & rho_mask));
          const uint32_t g = g_first;
          if (g >= lo && g + 3u <= hi) {
            *reinterpret_cast<float4 *>(row_ptr + g) = make_float4(s[0], s[1], s[2], s[3]);
          } else {
#pragma unroll
            for (uint32_t e = 0; e < 4; ++e)
              if (g + e >= lo && g + e <= hi) row_ptr[g + e] = s[e];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // This row ends in the NEXT tile: the rest of its last window is the head of the next row,
      // which that row's lane drew in ITS first window and handed over (lane 63's next row belongs
      // to another wave: that junction stays unmerged).  Written whole; the slot move below and
      // the next tile's own draws then overwrite the columns that are still this row's.
      const bool ends_next = merge && g_next / kTile == t + 1 && lane != 63;
      if (t + 3 >= n_tiles_w && ends_next) {
#pragma unroll
        for (uint32_t c = 0; c < kTile; ++c) logical[c] = next_head[c];
      }
      // slots [kTile, ...) are the next tile's slots [0, ...): the last group (from registers: all
      // of it is this row's), or whatever the general groups left there -- of which a row that ends
      // in the next tile keeps only its own columns (slot j is logical column j - kDraws + rho)
      if (interior_tile) {
#pragma unroll
        for (int j = 0; j < kDraws; ++j) my_row[j] = last[j];
      } else {
        const uint32_t keep = ends_next ? (g_next & (kTile - 1)) + kDraws - rho : static_cast<uint32_t>(kDraws);
        float up[kDraws];
#pragma unroll
        for (int j = 0; j < kDraws; ++j) up[j] = my_row[kTile + j];
#pragma unroll
        for (uint32_t j = 0; j < kDraws; ++j) {
          float *dst = j < keep ? my_row + j : my_trash;
          *dst = up[j];
        }
      }
    }
    if (active && k.d_final) k.d_final[i] = total;
  }
}

// ---- keepdata, comb form: every wave-wide store a set of WHOLE, unshared 128-byte lines ---------------
//
// keepdata_kernel above gives a wave 64 ADJACENT rows.  Adjacent rows share the line in which one ends
// and the next begins, their phases modulo a line all differ (per-lane delays, predicates, slot
// offsets), and a row of 361 values spans up to 392 columns of wave time.  Here a wave takes 64
// STREAMS of K consecutive rows whose first rows are 32 rows apart (stream l starts at row
// 2048 super + 32 l + K w): all 64 streams then start at the SAME offset phi inside a 128-byte line.
// Consequences:
//   * every piece of control is wave-uniform (SGPRs, scalar branches): the column cursor, the Philox
//     block index, row ends, tile flushes.  No lane is ever outside its stream, no predicate, no
//     per-lane delay: the draw and the step + ~2.5 VALU per stored value (20.2 in Gaussian mode; the tile
//     kernel with stream v2's draw: 39.8, this one: 33.0).
//   * a stream is stored from its first WHOLE line through the line in which its last row ends, and
//     it computes the head of the row that follows (the same counter stream: bit-identical to what
//     that row's own stream computes) to complete that line: every 128-byte line of the output is
//     written whole, once, by one wave -- also across waves and workgroups, where the tile kernel
//     leaves junction lines in two pieces.  tools/ubench_store_pattern.hip: this pattern streams at
//     6.1-6.4 TB/s, adjacent rows with shared lines at 3.8.  The extra work is <= 31 values per
//     stream (1.1 % at K = 4, P = 360).
//   * the LDS tile is column-major, [32 + D columns][64 lanes + 1]: a block's values go out as the
//     lane's own word of consecutive columns (conflict-free, no address arithmetic beyond one add per
//     block), the store phase reads 4 columns x 8 streams per 32 lanes (bank = column + stream:
//     conflict-free) and needs no per-lane state at all: stream r's line of tile t sits at
//     base + (r 32 L - phi + 32 t) floats, i.e. a scalar base plus one lane-constant offset.
// Needs n_periods to be a multiple of the draws per Philox block (a row is then whole blocks).
// Launch: rows [0, 2048 n_super) of the call; the host gives the remaining < 2048 rows to
// keepdata_kernel (whose first row's partial first line then repeats the head this kernel's last
// stream already wrote: same values).  d_final is filled by a separate gather (final_column_kernel).
constexpr int kCombMaxBlock = 1024;
constexpr int kCombColStride = 65;  // words between two columns of a tile: 64 lanes + 1
constexpr int comb_tile_words(int draws) { return (32 + draws) * kCombColStride; }

template <int kMode, bool kExactDiv, bool kDense, int kBlocksPerStep>
__global__ __launch_bounds__(kCombMaxBlock) void keepdata_comb_kernel(const KernelArgs k, const uint32_t rows_per_stream,
                                                                      const uint64_t n_wave_chunks, const uint64_t n_rows_total,
                                                                      unsigned long long *next_chunk) {
  constexpr int kDraws = Draws<kMode, kDense>::value;
  extern __shared__ __align__(16) unsigned char lds_raw[];
  float *lds_table = reinterpret_cast<float *>(lds_raw);
  const uint32_t table_words = is_table(kMode) ? k.table_len : bm_lds_words(kMode);
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = blockDim.x >> 6;
  float *tile = lds_table + table_words + wave * comb_tile_words(kDraws);
  stage_tables<kMode>(k, lds_table, blockDim.x);
  __syncthreads();
  const DrawRegs dr = make_draw_regs(k);

  const uint32_t K = rows_per_stream, waves_per_super = 32u / K;
  const uint32_t n_periods = k.n_periods, row_len = n_periods + 1u, n_blocks = n_periods / kDraws;
  const uint32_t sub = lane >> 3, quad = lane & 7u;
  const uint64_t base_f = reinterpret_cast<uintptr_t>(k.d_traj) >> 2;
  float *mine = tile + lane;                                           // column c of this lane's stream: mine[65 c]
  const float *take = tile + (4u * quad) * kCombColStride + sub;       // store phase: take[65 e + 8 it]
  const uint32_t lane_off = sub * (128u * row_len) + 16u * quad;       // bytes; row_len <= 2^20 (host)
  const uint64_t it_step = 1024ull * row_len;                          // bytes between two store iterations (8 streams)
  // One address register per store iteration, opaque to the compiler: with a single base and 32
  // immediate offsets it pairs the reads of two ITERATIONS into one ds_read2_b32 and then moves every
  // value into place for the 16-byte store (one v_mov per stored value); with a base of its own each
  // iteration pairs its own columns, and the pairs land in the store's registers.
  uint32_t take_off[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    take_off[it] = 8u * it;
    asm volatile("" : "+v"(take_off[it]));
  }

  // Wave chunks are handed out by a counter (*next_chunk, zeroed by the host before the launch): a
  // wave takes the next one when it is done.  A chunk is 64 K rows, a launch a few chunks per wave
  // (6 at 1.5e6 rows of 1001 values), so a fixed round-robin left whole waves idle for the last round
  // (1.71 ms where this takes 1.5x).  Every wave leaves the loop once the counter
  // has passed the end.
  (void)n_waves;
  for (;;) {
    unsigned long long taken = 0;
    if (lane == 0) taken = atomicAdd(next_chunk, 1ull);
    const uint64_t wc = (static_cast<uint64_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(taken >> 32))) << 32) |
                        __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(taken));
    if (wc >= n_wave_chunks) break;
    const uint64_t super = wc / waves_per_super;
    const uint32_t w = static_cast<uint32_t>(wc % waves_per_super);
    const uint64_t row0 = super * 2048ull + static_cast<uint64_t>(w) * K;  // first row of stream 0
    const uint32_t phi = static_cast<uint32_t>(base_f + row0 * row_len) & 31u;
    const uint32_t own = K * row_len;                                      // columns of the stream's own rows
    const uint32_t n_tiles = (phi + own + 31u) / 32u;                      // through the line its last row ends in
    // the row after stream r's last row exists for every r, unless this wave holds the call's last row
    const bool all_have_next = row0 + 32ull * 63ull + K < n_rows_total;
    uint64_t path = k.first_path + row0 + 32ull * lane;
    uint32_t path_lo = static_cast<uint32_t>(path), path_hi = static_cast<uint32_t>(path >> 32);
    char *line0 = reinterpret_cast<char *>(k.d_traj) + (static_cast<int64_t>(row0 * row_len) - phi) * 4;  // tile 0 of stream 0

    uint32_t col = phi, blk = 0, t = 0;
    float total = k.initial_capital;
    mine[col * kCombColStride] = total;  // values[0]
    col += 1;

    // Columns 0..31 of the tile are complete (one line of every stream): store them, move the columns
    // beyond down.  Returns true after the stream's last line.
    auto flush = [&]() -> bool {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      const bool first_partial = t == 0 && phi != 0;                 // streams start inside this line
      const bool last_tile = t + 1u == n_tiles;
      char *line_t = line0 + t * 128u;
      if (!first_partial && (!last_tile || all_have_next)) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const float *s = take + take_off[it];
          const float4 v = make_float4(s[0], s[kCombColStride], s[2 * kCombColStride], s[3 * kCombColStride]);
          *reinterpret_cast<float4 *>(line_t + it * it_step + lane_off) = v;
        }
      } else if (!first_partial || row0 == 0) {
        // the call's first line (stream 0 of the first wave: columns from phi on) or the line in
        // which the call's last row ends (no row follows: columns up to its last value)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const uint32_t r = sub + 8u * it;
          const float *s = take + 8 * it;
          uint32_t lo = 0u, hi = 31u;
          if (first_partial) {
            lo = phi;
            hi = (r == 0u) ? 31u : 0u;  // hi < lo: nothing (every other stream's first line belongs to the stream before it)
          }
          if (last_tile && !(row0 + 32ull * r + K < n_rows_total)) hi = (phi + own - 1u) & 31u;
#pragma unroll
          for (uint32_t e = 0; e < 4; ++e) {
            const uint32_t c = 4u * quad + e;
            if (c >= lo && c <= hi)
              *reinterpret_cast<float *>(line_t + it * it_step + lane_off + 4u * e) = s[e * kCombColStride];
          }
        }
      }
      if (last_tile) return true;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      // columns 32 .. col - 1 (at most kDraws of them) open the next tile
#pragma unroll
      for (int j = 0; j < kDraws; ++j) mine[j * kCombColStride] = mine[(32 + j) * kCombColStride];
      col -= 32u;
      t += 1;
      return false;
    };

    bool done = false;
    while (!done) {
      // kBlocksPerStep Philox blocks of the row are drawn together: independent instruction streams
      // for the scheduler to interleave (the kernel runs at under 4 waves per SIMD, LDS-limited);
      // their compounding chains follow one after the other.  The host guarantees that a row is a
      // whole number of steps.
      float a[kBlocksPerStep][kDraws];
      block_multipliers_multi<kMode, kDense, kBlocksPerStep, true>(k, dr, lds_table, path_lo, path_hi, blk, a);
#pragma unroll
      for (int b = 0; b < kBlocksPerStep; ++b) {
        if (done) break;
        float *dst = mine + col * kCombColStride;
#pragma unroll
        for (int j = 0; j < kDraws; ++j) {
          total = compound<kExactDiv>(total, a[b][j]);
          dst[j * kCombColStride] = total;
        }
        col += kDraws;
        if (b == kBlocksPerStep - 1) {
          blk += kBlocksPerStep;
          if (blk == n_blocks) {  // uniform: the next row (the stream's own, or the head that completes its last line)
            blk = 0;
            path += 1;
            path_lo = static_cast<uint32_t>(path);
            path_hi = static_cast<uint32_t>(path >> 32);
            total = k.initial_capital;
            mine[col * kCombColStride] = total;
            col += 1;
          }
        }
        if (col >= 32u) done = flush();  // uniform
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

// d_final[i] = last value of row i of a path-major trajectory array (the comb kernel's final values).
__global__ __launch_bounds__(kBlock) void final_column_kernel(const float *traj, uint64_t n_rows, uint32_t row_len, float *d_final) {
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n_rows;
       i += static_cast<uint64_t>(gridDim.x) * kBlock)
    d_final[i] = traj[i * row_len + (row_len - 1u)];
}

// Device self-test of the divide shortcut against the compiler's IEEE divide over the
// binary32 patterns [lo, hi): *count += mismatches of div100<false>(x) vs x / 100.0f.
__global__ __launch_bounds__(kBlock) void selftest_kernel(uint32_t lo, uint32_t hi,
                                                          unsigned long long *count) {
  unsigned long long bad = 0;
  const uint64_t n = static_cast<uint64_t>(hi) - lo;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<uint64_t>(gridDim.x) * kBlock) {
    const float x = __uint_as_float(lo + static_cast<uint32_t>(i));
    if (__float_as_uint(div100<false>(x)) != __float_as_uint(x / 100.0f)) ++bad;
  }
  bad = wave_sum(bad);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}

}  // namespace

hipError_t launch_selftest(uint32_t lo, uint32_t hi, unsigned long long *d_count, uint32_t grid,
                           hipStream_t stream) {
  hipLaunchKernelGGL(selftest_kernel, dim3(grid), dim3(kBlock), 0, stream, lo, hi, d_count);
  return hipGetLastError();
}

// `table_len` is 0 in Gaussian mode, where the Box-Muller tables take the table's place.
// `table_len` is 0 in Gaussian mode; `stream` (2 | 3) then picks the Box-Muller table set
static size_t draw_table_words(uint32_t table_len, int stream) {
  return table_len ? table_len : (stream == 2 ? kBmFloats : kBm3LdsWords);
}
size_t paths_lds_bytes(uint32_t table_len, uint32_t n_bins, int stream) {
  // [draw tables][histogram][pad to 8 bytes][red_scratch: 4 kWaves doubles][wave_part: kWaves BlockPartial]
  const size_t words = (draw_table_words(table_len, stream) + n_bins + 1u) & ~static_cast<size_t>(1);
  return words * 4u + 4u * kWaves * sizeof(double) + kWaves * sizeof(BlockPartial);
}
bool table_is_dense(uint32_t table_len);
// keepdata_kernel: tables + per wave 64 rows of (tile 16 | 32 columns + kDraws more slots, at the
// larger of the two strides) + one trash word per thread
size_t keepdata_lds_bytes(uint32_t table_len, int tile, int waves, int stream) {
  const size_t table_words = draw_table_words(table_len, stream);
  const int draws = (table_len && table_is_dense(table_len)) ? 8 : 4;
  const int row = keepdata_row_words(tile, draws);
  return (table_words + static_cast<size_t>(waves) * 64 * (row + 1)) * 4u;
}
// Static LDS of the kernels that read the v3 tables through absolute LDS addresses: must be 0.
hipError_t static_lds_bytes(size_t *bytes) {
  *bytes = 0;
  const void *kernels[] = {
      reinterpret_cast<const void *>(paths_kernel<SMMC_MODE_GAUSSIAN, kDivFast, false>),
      reinterpret_cast<const void *>(paths_kernel<SMMC_MODE_GAUSSIAN, kDivExact, false>),
      reinterpret_cast<const void *>(paths_kernel<SMMC_MODE_GAUSSIAN, kDivChecked, false>),
      reinterpret_cast<const void *>(keepdata_kernel<SMMC_MODE_GAUSSIAN, false, false, 32>),
      reinterpret_cast<const void *>(keepdata_kernel<SMMC_MODE_GAUSSIAN, true, false, 32>),
      reinterpret_cast<const void *>(keepdata_kernel<SMMC_MODE_GAUSSIAN, false, false, 16>),
      reinterpret_cast<const void *>(keepdata_kernel<SMMC_MODE_GAUSSIAN, true, false, 16>),
      reinterpret_cast<const void *>(keepdata_comb_kernel<SMMC_MODE_GAUSSIAN, false, false, 1>),
      reinterpret_cast<const void *>(keepdata_comb_kernel<SMMC_MODE_GAUSSIAN, true, false, 1>),
      reinterpret_cast<const void *>(keepdata_comb_kernel<SMMC_MODE_GAUSSIAN, false, false, 2>),
      reinterpret_cast<const void *>(keepdata_comb_kernel<SMMC_MODE_GAUSSIAN, true, false, 2>),
  };
  for (const void *kernel : kernels) {
    hipFuncAttributes attr;
    hipError_t err = hipFuncGetAttributes(&attr, kernel);
    if (err != hipSuccess) return err;
    if (attr.sharedSizeBytes > *bytes) *bytes = attr.sharedSizeBytes;
  }
  return hipSuccess;
}

size_t bm_tables_bytes(int stream) { return (stream == 2 ? kBmFloats : kBm3Floats) * sizeof(float); }

// Launches above 64 KiB of dynamic LDS (a 16384-entry table plus a large histogram) need
// the opt-in limit raised for that kernel; CDNA4 has 160 KiB per CU.
template <typename Kernel>
static hipError_t allow_lds(Kernel kernel, size_t lds) {
  if (lds <= 60u * 1024u) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             static_cast<int>(lds));
}

template <int kMode, int kDiv, bool kDense>
static hipError_t launch_paths_variant(const KernelArgs &a, uint32_t grid, size_t lds, hipStream_t stream) {
  hipError_t err = allow_lds(paths_kernel<kMode, kDiv, kDense>, lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL((paths_kernel<kMode, kDiv, kDense>), dim3(grid), dim3(kBlock), lds, stream, a);
  return hipGetLastError();
}

template <int kMode, bool kDense>
static hipError_t launch_paths_mode(const KernelArgs &a, int div, uint32_t grid, size_t lds, hipStream_t stream) {
  switch (div) {
    case SMMC_DIV_FAST: return launch_paths_variant<kMode, kDivFast, kDense>(a, grid, lds, stream);
    case SMMC_DIV_CHECKED: return launch_paths_variant<kMode, kDivChecked, kDense>(a, grid, lds, stream);
    default: return launch_paths_variant<kMode, kDivExact, kDense>(a, grid, lds, stream);
  }
}

bool table_is_dense(uint32_t table_len) { return table_len <= kDenseMaxTable; }

hipError_t launch_paths(const KernelArgs &a, int div, uint32_t grid, size_t lds_bytes, hipStream_t stream) {
  if (a.mode != SMMC_MODE_TABLE)
    return a.stream == 2 ? launch_paths_mode<kModeGaussianV2, false>(a, div, grid, lds_bytes, stream)
                         : launch_paths_mode<SMMC_MODE_GAUSSIAN, false>(a, div, grid, lds_bytes, stream);
  if (a.stream == 2)
    return table_is_dense(a.table_len) ? launch_paths_mode<kModeTableV2, true>(a, div, grid, lds_bytes, stream)
                                       : launch_paths_mode<kModeTableV2, false>(a, div, grid, lds_bytes, stream);
  return table_is_dense(a.table_len) ? launch_paths_mode<SMMC_MODE_TABLE, true>(a, div, grid, lds_bytes, stream)
                                     : launch_paths_mode<SMMC_MODE_TABLE, false>(a, div, grid, lds_bytes, stream);
}

hipError_t launch_finalize(const BlockPartial *partials, uint32_t n_partials, smmc_stats *d_stats,
                           uint32_t n_bins, hipStream_t stream, unsigned long long *hist_acc, uint32_t spread) {
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kFinalizeBlock), 0, stream, partials, n_partials, d_stats,
                     n_bins, hist_acc, spread);
  return hipGetLastError();
}

template <int kMode, bool kDense, int kTile>
static hipError_t launch_keepdata_tile(const KernelArgs &a, bool exact_div, uint32_t grid, uint32_t block,
                                        size_t lds, hipStream_t stream) {
  hipError_t err = exact_div ? allow_lds(keepdata_kernel<kMode, true, kDense, kTile>, lds)
                             : allow_lds(keepdata_kernel<kMode, false, kDense, kTile>, lds);
  if (err != hipSuccess) return err;
  if (exact_div)
    hipLaunchKernelGGL((keepdata_kernel<kMode, true, kDense, kTile>), dim3(grid), dim3(block), lds, stream, a);
  else
    hipLaunchKernelGGL((keepdata_kernel<kMode, false, kDense, kTile>), dim3(grid), dim3(block), lds, stream, a);
  return hipGetLastError();
}

template <int kMode, bool kDense>
static hipError_t launch_keepdata_mode(const KernelArgs &a, bool exact_div, int tile, int waves, uint32_t grid,
                                       hipStream_t stream) {
  const size_t lds = keepdata_lds_bytes(a.table_len, tile, waves, a.stream);
  const uint32_t block = static_cast<uint32_t>(waves) * 64u;
  if (waves < 1 || block > static_cast<uint32_t>(kKeepdataMaxBlock)) return hipErrorInvalidValue;
  switch (tile) {
    case 16: return launch_keepdata_tile<kMode, kDense, 16>(a, exact_div, grid, block, lds, stream);
    case 32: return launch_keepdata_tile<kMode, kDense, 32>(a, exact_div, grid, block, lds, stream);
    default: return hipErrorInvalidValue;
  }
}

size_t keepdata_comb_lds_bytes(uint32_t table_len, int waves, int stream) {
  const size_t table_words = draw_table_words(table_len, stream);
  const int draws = (table_len && table_is_dense(table_len)) ? 8 : 4;
  return (table_words + static_cast<size_t>(waves) * comb_tile_words(draws)) * 4u;
}
uint32_t keepdata_draws(uint32_t table_len) { return (table_len && table_is_dense(table_len)) ? 8u : 4u; }

template <int kMode, bool kDense, int kBlocksPerStep>
static hipError_t launch_comb_variant(const KernelArgs &a, bool exact_div, uint32_t rows_per_stream, uint64_t n_wave_chunks,
                                      uint64_t n_rows_total, int waves, uint32_t grid, unsigned long long *next_chunk,
                                      hipStream_t stream) {
  const size_t lds = keepdata_comb_lds_bytes(a.table_len, waves, a.stream);
  const uint32_t block = static_cast<uint32_t>(waves) * 64u;
  if (waves < 1 || block > static_cast<uint32_t>(kCombMaxBlock)) return hipErrorInvalidValue;
  hipError_t err = exact_div ? allow_lds(keepdata_comb_kernel<kMode, true, kDense, kBlocksPerStep>, lds)
                             : allow_lds(keepdata_comb_kernel<kMode, false, kDense, kBlocksPerStep>, lds);
  if (err != hipSuccess) return err;
  if (exact_div)
    hipLaunchKernelGGL((keepdata_comb_kernel<kMode, true, kDense, kBlocksPerStep>), dim3(grid), dim3(block), lds, stream, a,
                       rows_per_stream, n_wave_chunks, n_rows_total, next_chunk);
  else
    hipLaunchKernelGGL((keepdata_comb_kernel<kMode, false, kDense, kBlocksPerStep>), dim3(grid), dim3(block), lds, stream, a,
                       rows_per_stream, n_wave_chunks, n_rows_total, next_chunk);
  return hipGetLastError();
}

template <int kMode, bool kDense>
static hipError_t launch_comb_mode(const KernelArgs &a, bool exact_div, int blocks_per_step, uint32_t rows_per_stream,
                                   uint64_t n_wave_chunks, uint64_t n_rows_total, int waves, uint32_t grid,
                                   unsigned long long *next_chunk, hipStream_t stream) {
  return blocks_per_step == 2 ? launch_comb_variant<kMode, kDense, 2>(a, exact_div, rows_per_stream, n_wave_chunks, n_rows_total,
                                                                      waves, grid, next_chunk, stream)
                              : launch_comb_variant<kMode, kDense, 1>(a, exact_div, rows_per_stream, n_wave_chunks, n_rows_total,
                                                                      waves, grid, next_chunk, stream);
}

// blocks_per_step: 1 or 2 (2 needs n_periods / draws per block to be even)
hipError_t launch_keepdata_comb(const KernelArgs &a, bool exact_div, int blocks_per_step, uint32_t rows_per_stream,
                                uint64_t n_wave_chunks, uint64_t n_rows_total, int waves, uint32_t grid,
                                unsigned long long *next_chunk, hipStream_t stream) {
  hipError_t err = hipMemsetAsync(next_chunk, 0, sizeof(unsigned long long), stream);
  if (err != hipSuccess) return err;
  if (a.mode != SMMC_MODE_TABLE)
    return a.stream == 2 ? launch_comb_mode<kModeGaussianV2, false>(a, exact_div, blocks_per_step, rows_per_stream, n_wave_chunks,
                                                                    n_rows_total, waves, grid, next_chunk, stream)
                         : launch_comb_mode<SMMC_MODE_GAUSSIAN, false>(a, exact_div, blocks_per_step, rows_per_stream, n_wave_chunks,
                                                                       n_rows_total, waves, grid, next_chunk, stream);
  if (a.stream == 2)
    return table_is_dense(a.table_len)
               ? launch_comb_mode<kModeTableV2, true>(a, exact_div, blocks_per_step, rows_per_stream, n_wave_chunks, n_rows_total,
                                                      waves, grid, next_chunk, stream)
               : launch_comb_mode<kModeTableV2, false>(a, exact_div, blocks_per_step, rows_per_stream, n_wave_chunks, n_rows_total,
                                                       waves, grid, next_chunk, stream);
  return table_is_dense(a.table_len)
             ? launch_comb_mode<SMMC_MODE_TABLE, true>(a, exact_div, blocks_per_step, rows_per_stream, n_wave_chunks, n_rows_total,
                                                       waves, grid, next_chunk, stream)
             : launch_comb_mode<SMMC_MODE_TABLE, false>(a, exact_div, blocks_per_step, rows_per_stream, n_wave_chunks, n_rows_total,
                                                        waves, grid, next_chunk, stream);
}

hipError_t launch_final_column(const float *traj, uint64_t n_rows, uint32_t row_len, float *d_final, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(final_column_kernel, dim3(grid), dim3(kBlock), 0, stream, traj, n_rows, row_len, d_final);
  return hipGetLastError();
}

hipError_t launch_keepdata(const KernelArgs &a, bool exact_div, int tile, int waves, uint32_t grid,
                           hipStream_t stream) {
  if (a.mode != SMMC_MODE_TABLE)
    return a.stream == 2 ? launch_keepdata_mode<kModeGaussianV2, false>(a, exact_div, tile, waves, grid, stream)
                         : launch_keepdata_mode<SMMC_MODE_GAUSSIAN, false>(a, exact_div, tile, waves, grid, stream);
  if (a.stream == 2)
    return table_is_dense(a.table_len) ? launch_keepdata_mode<kModeTableV2, true>(a, exact_div, tile, waves, grid, stream)
                                       : launch_keepdata_mode<kModeTableV2, false>(a, exact_div, tile, waves, grid, stream);
  return table_is_dense(a.table_len) ? launch_keepdata_mode<SMMC_MODE_TABLE, true>(a, exact_div, tile, waves, grid, stream)
                                     : launch_keepdata_mode<SMMC_MODE_TABLE, false>(a, exact_div, tile, waves, grid, stream);
}

}  // namespace smmc
