/*
 * smmc_oracle.c -- CPU restatement of the Monte-Carlo returns engine.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
 * stock_market_monte_carlo_amd/, include/, the CLIs) may include, link, import
 * or execute this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the timed CPU
 * baseline.
 *
 * PINNING STATUS: "parity unpinned" against an executed reference.
 *   The reference ships no tests, golden vectors or fixtures for this path, and
 *   its translation unit src/simulations.cpp cannot be compiled in this image
 *   without writing stand-ins for third-party headers it includes ("csv.h" from
 *   fast-cpp-csv-parser is absent), so no reference binary was built or run.
 *   What IS pinned:
 *     - mt19937 + uniform_int_distribution<int> (the libstdc++ code the reference
 *       calls at src/simulations.cpp:245-247) against the system libstdc++ 11.4
 *       through oracle/pin/pin_libstdcxx.cpp -> tests/golden/libstdcxx_*.json,
 *       and the ISO C++ known answer (10000th output of mt19937() == 4123659995);
 *       the same fixture holds whole reference-style paths at the device kernels' edge
 *       lengths, paths in which the library REJECTS a generator output, and whole
 *       keepdata trajectories -- engine (R) below reproduces all of them, and since
 *       round 3 so do the HIP kernels of the reference's own stream
 *       (SMMC_FLAG_STREAM_REF; tests/test_ref_stream_gpu.py compares them with the
 *       fixture directly, not through this file);
 *     - Philox4x32-10 against the published Random123 known-answer vectors;
 *     - the statistics record (sum, mean, below-count) against update_mean_std /
 *       update_count_below_min compiled from the reference's examples/benchmark_mc_gpu.cpp
 *       (oracle/Makefile target _ref; tests/test_ref_callers_cpu.py).
 *   The compounding arithmetic itself is three IEEE-754 binary32 operations
 *   restated from src/simulations.cpp:14-16.
 *
 * Two engines live here:
 *   (R) "reference-faithful": per-path mt19937, Lemire index map, update_fund --
 *       follows src/simulations.cpp:204-266 with deterministic per-path seeds
 *       (the reference seeds from std::random_device and has no seed argument).
 *       This is what bench.py times as the CPU baseline ("port").
 *   (C) "counter stream v3" (default) and v2 (orc_params.stream = 2): Philox4x32-10 keyed by
 *       the 64-bit seed.  The stream number selects the COUNTER LAYOUT IN BOTH MODES --
 *       v3: (step block, path id lo, path id hi, mode tag); v2: (path id lo, path id hi,
 *       step block, mode tag) -- and the Gaussian draw (v3's cheaper Box-Muller, or
 *       v2's).  So table-mode results for a given seed differ between the streams too:
 *       round 1's table-mode values are reproduced only with stream = 2.  Table-indexed
 *       draws (eight per block for tables <= 2048 entries) or Box-Muller Gaussian draws;
 *       the same three-rounding compounding step.  The HIP kernels must reproduce this
 *       engine bit-for-bit (final values, histogram bucket counts, below-threshold counts).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: every fused multiply-add below is an explicit
 * fmaf(); nothing else may be contracted.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Compounding core: src/simulations.cpp:14-22                                */
/* ------------------------------------------------------------------------- */

/* src/simulations.cpp:14-16 -- fund * (100.0f + r) / 100, r in percent.
 * Three binary32 roundings: add, multiply, true divide (int 100 -> 100.0f). */
ORC_API float orc_update_fund(float fund_value, float period_return) {
  float a = 100.0f + period_return;
  float m = fund_value * a;
  return m / 100.0f;
}

/* src/simulations.cpp:18-22 -- totals[0] is the caller's start value;
 * totals has n_periods + 1 entries. */
ORC_API void orc_many_updates(const float *returns, float *totals, uint32_t n_periods) {
  for (uint32_t i = 0; i < n_periods; i++)
    totals[i + 1] = orc_update_fund(totals[i], returns[i]);
}

/* ------------------------------------------------------------------------- */
/* mt19937 (ISO C++ [rand.eng.mers], the engine at src/simulations.cpp:246)   */
/* ------------------------------------------------------------------------- */

typedef struct {
  uint32_t mt[624];
  int idx;
} orc_mt19937;

static void mt_seed(orc_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; i++)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

static void mt_twist(orc_mt19937 *g) {
  uint32_t *mt = g->mt;
  for (int k = 0; k < 624; k++) {
    uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
    uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
    if (y & 1u) v ^= 0x9908b0dfu;
    mt[k] = v;
  }
  g->idx = 0;
}

static inline uint32_t mt_next(orc_mt19937 *g) {
  if (g->idx >= 624) mt_twist(g);
  uint32_t y = g->mt[g->idx++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* libstdc++ 11 uniform_int_distribution<int>(0, range-1)(mt19937):
 * /usr/include/c++/11/bits/uniform_int_dist.h:241-268 (_S_nd, Lemire's nearly
 * divisionless method with 64-bit product) selected at :294-323 because
 * mt19937's range is exactly 2^32-1.  Call site: src/simulations.cpp:247,250. */
static inline uint32_t lemire_index(orc_mt19937 *g, uint32_t range) {
  uint64_t product = (uint64_t)mt_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    uint32_t threshold = (uint32_t)(-range) % range;
    while (low < threshold) {
      product = (uint64_t)mt_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32);
}

/* n raw outputs of mt19937(seed) -- for the ISO known-answer test. */
ORC_API void orc_mt19937_raw(uint32_t seed, uint32_t n, uint32_t *out) {
  orc_mt19937 g;
  mt_seed(&g, seed);
  for (uint32_t i = 0; i < n; i++) out[i] = mt_next(&g);
}

/* n draws of uniform_int_distribution<int>(0, range-1) on mt19937(seed). */
ORC_API void orc_mt19937_indices(uint32_t seed, uint32_t range, uint32_t n, uint32_t *out) {
  orc_mt19937 g;
  mt_seed(&g, seed);
  for (uint32_t i = 0; i < n; i++) out[i] = lemire_index(&g, range);
}

/* (R) one path of src/simulations.cpp:240-252 with an explicit seed. */
static float ref_one_path(uint32_t seed, uint32_t n_periods, float initial_capital,
                          const float *table, uint32_t table_len) {
  orc_mt19937 g;
  mt_seed(&g, seed);
  float total = initial_capital;
  for (uint32_t i = 0; i < n_periods; i++)
    total = orc_update_fund(total, table[lemire_index(&g, table_len)]);
  return total;
}

/* (R) src/simulations.cpp:204-266: OpenMP schedule(dynamic) over blocks of 1000
 * paths.  Path `id` is seeded with (uint32_t)(seed0 + id) where the reference
 * uses a fresh std::random_device per path.  n_threads <= 0 means the
 * reference's max(1, hardware_concurrency - 1) (:218-219).  Returns the number
 * of threads used.  final_values must hold n_paths floats (:252). */
ORC_API int orc_ref_mc_simulations(int64_t n_paths, uint32_t n_periods, float initial_capital,
                                   const float *table, uint32_t table_len, uint32_t seed0,
                                   float *final_values, int n_threads) {
  const int64_t block_size = 1000;
  const int64_t n_blocks = (n_paths + block_size - 1) / block_size;
  int used = 1;
#ifdef _OPENMP
  if (n_threads <= 0) {
    n_threads = omp_get_num_procs() - 1;
    if (n_threads < 1) n_threads = 1;
  }
  used = n_threads;
#pragma omp parallel for schedule(dynamic) num_threads(n_threads)
#endif
  for (int64_t b = 0; b < n_blocks; b++) {
    int64_t first = b * block_size;
    int64_t last = first + block_size < n_paths ? first + block_size : n_paths;
    for (int64_t id = first; id < last; id++)
      final_values[id] =
          ref_one_path((uint32_t)(seed0 + (uint64_t)id), n_periods, initial_capital, table, table_len);
  }
  return used;
}

/* ------------------------------------------------------------------------- */
/* (C) counter stream v2                                                      */
/* ------------------------------------------------------------------------- */

/* Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11; Random123 philox.h).  Published algorithm restated. */
ORC_API void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define ORC_MODE_TABLE 0
#define ORC_MODE_GAUSSIAN 1

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* Box-Muller tables of counter stream v2 (generated by tools/gen_bm_tables.py; the HIP
 * kernels compile the identical numbers). */
#include "smmc_bm_tables.inc"

/* Radius r = sqrt(-2 ln U) without log or sqrt.  w = ua ^ (ua >>a 31) is the distance of the
 * uniform from the end of (0,1) it is nearer to (side 0: U < 1/2, measured from 0; side 1:
 * from 1); f = (float)(2 w + 1) rounds it to binary32, and U (side 0) or 1 - U (side 1) is
 * f / 2^33.  The exponent of f picks one of 33 octaves, its top four mantissa bits one of 16
 * sub-intervals, the remaining 19 bits x in [-0.5, 0.5); one cubic per bin, three fmaf.
 * An integer-to-float conversion, integer steps and single IEEE operations: bit-reproducible. */
ORC_API float orc_bm_radius(uint32_t ua) {
  uint32_t mask = (uint32_t)((int32_t)ua >> 31); /* all ones when U >= 1/2 */
  uint32_t w1 = ((ua ^ mask) << 1) | 1u;         /* 2 w + 1: odd, in [1, 2^32) */
  uint32_t bits = f2u((float)w1);                /* round to nearest even, exponent 127 .. 159 */
  uint32_t entry = (mask & 528u) + (bits >> 19) - (127u << 4); /* 528 = 33 octaves x 16 */
  float x = u2f(0x3f800000u | ((bits << 4) & 0x007ffff0u)) - 1.5f;
  const float *k = smmc_bm_radius[entry];
  return fmaf(fmaf(fmaf(k[3], x, k[2]), x, k[1]), x, k[0]);
}

/* Box-Muller on two 32-bit words, scaled: ua -> radius (above), ub -> angle theta = 2 pi ub / 2^32
 * = 2 pi i / 256 + delta with i = (ub + 2^23) >> 24 and delta = sext24(ub) * 2 pi / 2^32,
 * |delta| <= pi/256: (cos, sin)(2 pi i / 256) from the table, rotated by
 * sin(delta) = delta (1 - delta^2/6), cos(delta) = 1 - delta^2/2 (errors < 1e-9).
 * Returns the two draws  fma(r * scale, cos(theta), shift)  and  fma(r * scale, sin(theta), shift):
 * N(shift, scale) variates; scale = 1, shift = 0 gives the standard normals themselves. */
ORC_API void orc_box_muller_scaled(uint32_t ua, uint32_t ub, float scale, float shift, float *d_cos, float *d_sin) {
  float rs = orc_bm_radius(ua) * scale;
  uint32_t i = (ub + 0x00800000u) >> 24;
  int32_t d = (int32_t)(ub << 8) >> 8; /* low 24 bits, sign-extended */
  float delta = (float)d * 0x1.921fb6p-30f;
  float d2 = delta * delta;
  float sd = delta * fmaf(d2, -0x1.555556p-3f, 1.0f);
  float cd = fmaf(d2, -0.5f, 1.0f);
  float ci = smmc_bm_trig[i][0], si = smmc_bm_trig[i][1];
  float ct = fmaf(-si, sd, ci * cd);
  float st = fmaf(ci, sd, si * cd);
  *d_cos = fmaf(rs, ct, shift);
  *d_sin = fmaf(rs, st, shift);
}

ORC_API void orc_box_muller(uint32_t ua, uint32_t ub, float *z_cos, float *z_sin) {
  orc_box_muller_scaled(ua, ub, 1.0f, 0.0f, z_cos, z_sin);
}

/* ---- counter stream v3: the same construction with cheaper arithmetic ---------------------------
 * (round 2: the Gaussian path step is bound by VALU instruction count; v2 needs 30.5 per period on
 * the device, v3 17.5.  The table DRAW -- index digits, table entry, step -- is the same in v2 and v3;
 * the Philox counter layout is not (path_returns_block), so table-mode VALUES differ between them.)
 * Radius: the first word read as int32 is the SIGNED distance d of the uniform from the nearer end of
 * (0, 1) in units of 2^-32 (d > 0: from 0; d < 0: from 1); f = (float)d (round to nearest even,
 * |f| <= 2^31), u = |f| / 2^32 in (0, 1/2]; d = 0 stands for u = 2^-33.  Exponent 127 .. 158: 32 octaves x
 * 2^SUB_BITS sub-intervals per side, stored rotated so that the table index is ONE bit-field of f's
 * pattern, the side its sign bit.  The cubic is in f itself: the table's coefficients carry the bin's
 * position, its octave's powers of two and (odd ones) the sign.  Tail: 6.76 sigma at d = 0, 6.66 at 1. */
#define BM3_LOW_BITS (23 - SMMC_BM3_SUB_BITS)
#define BM3_SIDE_ENTRIES (32u << SMMC_BM3_SUB_BITS)
/* r * scale as the kernels form it: every coefficient times scale, rounded once (the device does that
 * while it stages the table in LDS), then the cubic -- the draw needs no multiply of its own.
 * scale = 1 is the radius itself. */
ORC_API float orc_bm3_radius_scaled(uint32_t ua, float scale) {
  float f = (float)(int32_t)ua;
  uint32_t bits = f2u(f);                       /* 0, or exponent 127 .. 158 with the side as the sign */
  uint32_t entry = ((bits >> BM3_LOW_BITS) & (BM3_SIDE_ENTRIES - 1u)) | ((bits >> 31) ? BM3_SIDE_ENTRIES : 0u);
  const float *k = smmc_bm3_radius[entry];
  float k0 = k[0] * scale, k1 = k[1] * scale, k2 = k[2] * scale, k3 = k[3] * scale;
  return fmaf(fmaf(fmaf(k3, f, k2), f, k1), f, k0);
}

ORC_API float orc_bm3_radius(uint32_t ua) { return orc_bm3_radius_scaled(ua, 1.0f); }

/* Angle: theta = 2 pi (ub mod 2^ANGLE_BITS) / 2^ANGLE_BITS (ANGLE_BITS = 30: the word's top two bits
 * are not used) = theta_i + delta, theta_i the MIDDLE of sector i = the TRIG_BITS bits below bit 30
 * (no rounding add; on the device sector * 8 is one masked read of the word's upper half) and delta =
 * (low 19 bits - half a sector) 2 pi / 2^30, |delta| <= dmax = pi / 2^TRIG_BITS, formed WITHOUT an
 * integer-to-float conversion: the low bits are OR-ed into the mantissa of 1.0f and one fma scales and
 * centres them.  The table's (cos, sin) are rotated by delta to FIRST order, (c - s delta, s + c delta):
 * a vector at angle theta_i + atan(delta) (off by delta^3/3 <= 1.2e-9) of length sqrt(1 + delta^2); the
 * table carries kappa = 1/sqrt(1 + dmax^2/3), which makes the mean square length 1, so a draw is
 * r cos(theta) times a factor within -3.9e-7 .. +7.8e-7 of 1 that depends on where in its sector the
 * angle falls (2048 sectors).  Against a second-order rotation this saves 4 instructions per pair; the
 * factor is below the binary32 rounding of the draw itself (tests/test_numerics_cpu.py states the
 * bounds). */
#define BM3_RES_BITS (SMMC_BM3_ANGLE_BITS - SMMC_BM3_TRIG_BITS)
ORC_API void orc_box_muller3_scaled(uint32_t ua, uint32_t ub, float scale, float shift, float *d_cos, float *d_sin) {
  float rs = orc_bm3_radius_scaled(ua, scale);
  uint32_t i = (ub >> BM3_RES_BITS) & ((1u << SMMC_BM3_TRIG_BITS) - 1u);
  float y = u2f(0x3f800000u | (ub & ((1u << BM3_RES_BITS) - 1u)));
  float delta = fmaf(y, SMMC_BM3_ANGLE_K, -SMMC_BM3_ANGLE_C);
  float ci = smmc_bm3_trig[i][0], si = smmc_bm3_trig[i][1];
  float ct = fmaf(-si, delta, ci);
  float st = fmaf(ci, delta, si);
  *d_cos = fmaf(rs, ct, shift);
  *d_sin = fmaf(rs, st, shift);
}

ORC_API void orc_box_muller3(uint32_t ua, uint32_t ub, float *z_cos, float *z_sin) {
  orc_box_muller3_scaled(ua, ub, 1.0f, 0.0f, z_cos, z_sin);
}

/* max |r - sqrt(-2 ln u)| over ua = lo, lo + stride, ... < hi (v3 radius against double) */
ORC_API double orc_bm3_radius_scan(uint64_t lo, uint64_t hi, uint64_t stride) {
  double worst = 0.0;
  for (uint64_t a = lo; a < hi; a += stride) {
    uint32_t ua = (uint32_t)a;
    double f = (double)(float)(int32_t)ua;
    double u = ua ? fabs(f) / 4294967296.0 : 0x1p-33;
    double want = f < 0 ? sqrt(-2.0 * log1p(-u)) : sqrt(-2.0 * log(u));
    double e = fabs((double)orc_bm3_radius(ua) - want);
    if (e > worst) worst = e;
  }
  return worst;
}

typedef struct {
  int32_t mode;          /* ORC_MODE_* */
  uint32_t n_periods;
  uint64_t seed;         /* Philox key = (lo32, hi32) */
  uint64_t first_path;   /* global id of path 0 of this call */
  uint64_t n_paths;
  float initial_capital;
  float gauss_mean;      /* percent per period */
  float gauss_std;       /* percent per period */
  const float *table;    /* percent per period, table_len entries */
  uint32_t table_len;
  uint32_t n_bins;       /* 0 = no histogram */
  float hist_lo, hist_hi;
  float below_threshold;
  uint32_t stream;       /* 2 = counter stream v2, anything else (0, 3) = v3: the Philox counter layout in BOTH modes
                          * (path_returns_block) and the Gaussian draw */
} orc_params;

typedef struct {
  uint64_t count;
  uint64_t below;        /* v < below_threshold */
  uint64_t underflow;    /* v < hist_lo */
  uint64_t overflow;     /* !(v < hist_hi), NaN included */
  double sum;
  double sumsq;
  float min, max;
} orc_stats;

/* Table-draw schedule.  Tables of up to ORC_DENSE_MAX_TABLE entries take EIGHT indices
 * from one Philox block ("dense"): each 64-bit half (u0:u1), (u2:u3) is a fraction x in
 * [0,1); digit k = floor(T * frac(T^k * x)) for k = 0..2 by exact 64x32-bit multiplies,
 * and the fourth digit from the top 32 bits of what is left.  Relative bias of digit k is
 * below T^(k+1) / 2^64 (k < 3) and T^4/2^64 + T/2^32 (k = 3): < 1.3e-6 for T <= 2048.
 * Larger tables take one index per 32-bit word ("sparse", four per block). */
#define ORC_DENSE_MAX_TABLE 2048u

ORC_API uint32_t orc_draws_per_block(int32_t mode, uint32_t table_len) {
  return (mode == ORC_MODE_TABLE && table_len <= ORC_DENSE_MAX_TABLE) ? 8u : 4u;
}

static void digits4(uint32_t h, uint32_t l, uint32_t T, uint32_t idx[4]) {
  for (int d = 0; d < 3; d++) {
    uint64_t pl = (uint64_t)l * T;
    uint64_t ph = (uint64_t)h * T + (pl >> 32);
    idx[d] = (uint32_t)(ph >> 32);
    h = (uint32_t)ph;
    l = (uint32_t)pl;
  }
  idx[3] = (uint32_t)(((uint64_t)h * T) >> 32);
}

/* The draws of Philox block `blk` of one path: orc_draws_per_block() of them.  out[] are the period
 * returns in percent, mult[] the multipliers a = 100 + return the compounding step uses (period p
 * uses draw p % D of block p / D).  Table mode and Gaussian v2: the return is drawn (table entry;
 * fma(std, z, mean)) and a = 100.0f + return.  Gaussian v3: the MULTIPLIER is drawn,
 * a = fma(r std, cos theta, 100.0f + mean), and the return is defined as a - 100.0f (exact for
 * a in [50, 200], where update_fund(total, return) then reproduces total * a / 100 bit for bit). */
static void path_returns_block(const orc_params *p, uint64_t path, uint32_t blk, float out[8],
                               uint32_t idx_out[8], float mult[8]) {
  /* counter: stream v3 counts blocks in the FIRST word, (block, path_lo, path_hi, mode) -- on the device
   * the first two Philox rounds then cost two instructions instead of four --, stream v2 in the third */
  uint32_t ctr[4] = {(uint32_t)path, (uint32_t)(path >> 32), blk, (uint32_t)p->mode};
  if (p->stream != 2) {
    ctr[0] = blk;
    ctr[1] = (uint32_t)path;
    ctr[2] = (uint32_t)(path >> 32);
  }
  uint32_t key[2] = {(uint32_t)p->seed, (uint32_t)(p->seed >> 32)};
  uint32_t u[4];
  float a[8];
  uint32_t n = orc_draws_per_block(p->mode, p->table_len);
  orc_philox4x32_10(ctr, key, u);
  if (p->mode == ORC_MODE_TABLE) {
    uint32_t idx[8];
    if (n == 8) {
      digits4(u[0], u[1], p->table_len, idx);
      digits4(u[2], u[3], p->table_len, idx + 4);
    } else {
      for (int j = 0; j < 4; j++) idx[j] = (uint32_t)(((uint64_t)u[j] * p->table_len) >> 32);
    }
    for (uint32_t j = 0; j < n; j++) {
      if (idx_out) idx_out[j] = idx[j];
      out[j] = p->table[idx[j]];
      a[j] = 100.0f + out[j];
    }
  } else if (p->stream == 2) {
    orc_box_muller_scaled(u[0], u[1], p->gauss_std, p->gauss_mean, &out[0], &out[1]);
    orc_box_muller_scaled(u[2], u[3], p->gauss_std, p->gauss_mean, &out[2], &out[3]);
    for (int j = 0; j < 4; j++) a[j] = 100.0f + out[j];
  } else {
    const float shift = 100.0f + p->gauss_mean;
    orc_box_muller3_scaled(u[0], u[1], p->gauss_std, shift, &a[0], &a[1]);
    orc_box_muller3_scaled(u[2], u[3], p->gauss_std, shift, &a[2], &a[3]);
    for (int j = 0; j < 4; j++) out[j] = a[j] - 100.0f;
  }
  if (mult)
    for (uint32_t j = 0; j < n; j++) mult[j] = a[j];
}

/* Writes the n_periods returns of global path `path` (percent). */
ORC_API void orc_counter_path_returns(const orc_params *p, uint64_t path, float *returns) {
  const uint32_t D = orc_draws_per_block(p->mode, p->table_len);
  for (uint32_t i = 0; i < p->n_periods; i += D) {
    float r[8];
    path_returns_block(p, path, i / D, r, 0, 0);
    for (uint32_t j = 0; j < D && i + j < p->n_periods; j++) returns[i + j] = r[j];
  }
}

/* Table indices drawn by a path (table mode), for distribution tests. */
ORC_API void orc_counter_path_indices(const orc_params *p, uint64_t path, uint32_t *indices) {
  const uint32_t D = orc_draws_per_block(p->mode, p->table_len);
  for (uint32_t i = 0; i < p->n_periods; i += D) {
    float r[8];
    uint32_t idx[8];
    path_returns_block(p, path, i / D, r, idx, 0);
    for (uint32_t j = 0; j < D && i + j < p->n_periods; j++) indices[i + j] = idx[j];
  }
}

static float counter_one_path(const orc_params *p, uint64_t path, float *trajectory) {
  const uint32_t D = orc_draws_per_block(p->mode, p->table_len);
  float total = p->initial_capital;
  if (trajectory) trajectory[0] = total;
  for (uint32_t i = 0; i < p->n_periods; i += D) {
    float r[8], a[8];
    path_returns_block(p, path, i / D, r, 0, a);
    for (uint32_t j = 0; j < D && i + j < p->n_periods; j++) {
      /* update_fund (src/simulations.cpp:14-16) from its second rounding on: a = 100.0f + r is formed above */
      const float m = total * a[j];
      total = m / 100.0f;
      if (trajectory) trajectory[i + j + 1] = total;
    }
  }
  return total;
}

/* Histogram bucket contract (build-defined; the reference bins only inside a
 * third-party plotting call): inv = (double)n_bins / ((double)hi - (double)lo);
 * v < lo -> underflow; v < hi -> bucket min((int)(((double)v - lo) * inv),
 * n_bins - 1); anything else (v >= hi or NaN) -> overflow. */
ORC_API int32_t orc_hist_bucket(float v, float lo, float hi, uint32_t n_bins) {
  double inv = (double)n_bins / ((double)hi - (double)lo);
  if (v < lo) return -1;
  if (v < hi) {
    int32_t b = (int32_t)(((double)v - (double)lo) * inv);
    return b < (int32_t)n_bins - 1 ? b : (int32_t)n_bins - 1;
  }
  return (int32_t)n_bins;
}

/* (C) engine.  Any of final_values (n_paths floats), hist (n_bins u64, zeroed
 * here), stats, trajectories (n_paths x (n_periods+1) floats, path-major) may be
 * NULL.  Parallelised over paths when OpenMP is on; sums are accumulated in
 * global path order afterwards so the result does not depend on the thread
 * count. */
ORC_API int orc_counter_mc(const orc_params *p, float *final_values, uint64_t *hist, orc_stats *stats,
                           float *trajectories, int n_threads) {
  if (p->mode == ORC_MODE_TABLE && (p->table == 0 || p->table_len == 0)) return -1;
  int64_t n = (int64_t)p->n_paths;
  float *fv = final_values;
  if (!fv) {
    fv = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    if (!fv) return -2;
  }
#ifdef _OPENMP
  if (n_threads <= 0) n_threads = omp_get_num_procs();
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
  for (int64_t i = 0; i < n; i++) {
    float *traj = trajectories ? trajectories + (size_t)i * (p->n_periods + 1) : 0;
    fv[i] = counter_one_path(p, p->first_path + (uint64_t)i, traj);
  }
  if (hist) memset(hist, 0, sizeof(uint64_t) * p->n_bins);
  orc_stats s;
  memset(&s, 0, sizeof s);
  s.min = INFINITY;
  s.max = -INFINITY;
  for (int64_t i = 0; i < n; i++) {
    float v = fv[i];
    s.count++;
    if (v < p->below_threshold) s.below++;
    s.sum += (double)v;
    s.sumsq += (double)v * (double)v;
    if (v < s.min) s.min = v;
    if (v > s.max) s.max = v;
    if (p->n_bins) {
      int32_t b = orc_hist_bucket(v, p->hist_lo, p->hist_hi, p->n_bins);
      if (b < 0) s.underflow++;
      else if (b >= (int32_t)p->n_bins) s.overflow++;
      else if (hist) hist[b]++;
    }
  }
  if (stats) *stats = s;
  if (!final_values) free(fv);
  return 0;
}

/* Per-chunk (256 consecutive paths) mean and population variance -- the shape of
 * the reference's reduceBlock output (src/simulations.cu:231-246, one pair per
 * 256-thread block) but dividing by the true chunk length for a partial last
 * chunk.  Double accumulation, rounded to float once. */
ORC_API void orc_chunk_mean_var(const float *values, uint64_t n, uint32_t chunk, float *means,
                                float *variances) {
  uint64_t n_chunks = (n + chunk - 1) / chunk;
  for (uint64_t c = 0; c < n_chunks; c++) {
    uint64_t a = c * chunk, b = a + chunk < n ? a + chunk : n;
    double s = 0;
    for (uint64_t i = a; i < b; i++) s += (double)values[i];
    double m = s / (double)(b - a);
    double q = 0;
    for (uint64_t i = a; i < b; i++) {
      double d = (double)values[i] - m;
      q += d * d;
    }
    means[c] = (float)m;
    variances[c] = (float)(q / (double)(b - a));
  }
}

/* ------------------------------------------------------------------------- */
/* Checks used by tests for device-side shortcuts                             */
/* ------------------------------------------------------------------------- */

/* The HIP kernels divide by 100 with  fma(x, ch, fl(x * cl)),  ch = fl(1/100), cl = fl(1/100 - ch)
 * (two instructions; Brisebarre-Muller multiplication by a two-word constant).  Counts the binary32
 * patterns in [bits_lo, bits_hi) for which that differs from the IEEE quotient x / 100.0f; the first
 * mismatch is stored.  form = 3 counts round 1's three-instruction form instead
 * (q = x ch; e = fma(-100, q, x); fma(e, ch, q)). */
ORC_API uint64_t orc_div100_mismatches_form(uint32_t bits_lo, uint32_t bits_hi, uint32_t *first_bad, int form) {
  const float ch = 0.01f, cl = 0x1.eb851ep-33f;
  uint64_t bad = 0;
  for (uint64_t b = bits_lo; b < bits_hi; b++) {
    float x = u2f((uint32_t)b);
    float got;
    if (form == 3) {
      float q = x * ch;
      float e = fmaf(-100.0f, q, x);
      got = fmaf(e, ch, q);
    } else {
      got = fmaf(x, ch, x * cl);
    }
    float ref = x / 100.0f;
    if (f2u(got) != f2u(ref)) {
      if (!bad && first_bad) *first_bad = (uint32_t)b;
      bad++;
    }
  }
  return bad;
}

ORC_API uint64_t orc_div100_mismatches(uint32_t bits_lo, uint32_t bits_hi, uint32_t *first_bad) {
  return orc_div100_mismatches_form(bits_lo, bits_hi, first_bad, 2);
}

/* cl as the kernels spell it: fl(1/100 - fl(1/100)) */
ORC_API float orc_div100_cl(void) { return (float)(0.01L - (long double)0.01f); }

/* Scans ua = lo, lo + stride, ... < hi: largest absolute error of orc_bm_radius against
 * sqrt(-2 ln((2 ua + 1) / 2^33)) in double precision. */
ORC_API double orc_bm_radius_scan(uint64_t lo, uint64_t hi, uint64_t stride) {
  double worst = 0;
  for (uint64_t a = lo; a < hi; a += stride) {
    /* the radius is defined on the ROUNDED distance fl(2 w + 1) */
    uint32_t wv = a < 0x80000000ull ? (uint32_t)a : (uint32_t)(0xFFFFFFFFull - a);
    double u = (double)(float)((wv << 1) | 1u) * 0x1p-33;
    double ref = a < 0x80000000ull ? sqrt(-2.0 * log(u)) : sqrt(-2.0 * log1p(-u));
    double err = fabs((double)orc_bm_radius((uint32_t)a) - ref);
    if (err > worst) worst = err;
  }
  return worst;
}

/* ------------------------------------------------------------------------- */
/* Statistics of an array of values (the callers' passes over final_values)   */
/* ------------------------------------------------------------------------- */

/* Same record as orc_counter_mc computes, for an arbitrary array: sequential double sums
 * (examples/visualize_returns_cpu_v2.cpp:113-138, examples/benchmark_mc_gpu.cpp:7-41). */
ORC_API void orc_values_stats(const float *v, uint64_t n, float below_threshold, uint32_t n_bins, float lo,
                              float hi, orc_stats *out, uint64_t *hist) {
  orc_stats s;
  memset(&s, 0, sizeof s);
  s.min = INFINITY;
  s.max = -INFINITY;
  if (hist) memset(hist, 0, sizeof(uint64_t) * n_bins);
  for (uint64_t i = 0; i < n; i++) {
    float x = v[i];
    s.count++;
    if (x < below_threshold) s.below++;
    s.sum += (double)x;
    s.sumsq += (double)x * (double)x;
    if (x < s.min) s.min = x;
    if (x > s.max) s.max = x;
    if (n_bins) {
      int32_t b = orc_hist_bucket(x, lo, hi, n_bins);
      if (b < 0) s.underflow++;
      else if (b >= (int32_t)n_bins) s.overflow++;
      else if (hist) hist[b]++;
    }
  }
  *out = s;
}

static int cmp_float(const void *a, const void *b) {
  float x = *(const float *)a, y = *(const float *)b;
  return (x > y) - (x < y);
}

/* k-th smallest (0-based) by fully sorting a copy -- what examples/visualize_returns_gpu.cpp:
 * 108-109 does with std::sort (and :83-111 of the cpu_v2 twin with std::nth_element). */
ORC_API int orc_order_statistics(const float *v, uint64_t n, const uint64_t *ranks, uint32_t n_ranks, float *out) {
  for (uint32_t q = 0; q < n_ranks; q++)
    if (ranks[q] >= n) return -1;  /* found by the ASan driver (oracle/sanitize): was an out-of-bounds read */
  float *copy = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1));
  if (!copy) return -2;
  memcpy(copy, v, sizeof(float) * (size_t)n);
  qsort(copy, (size_t)n, sizeof(float), cmp_float);
  for (uint32_t q = 0; q < n_ranks; q++) out[q] = copy[ranks[q]];
  free(copy);
  return 0;
}
