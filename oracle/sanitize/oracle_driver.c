/* oracle_driver.c -- drives every entry point of oracle/smmc_oracle.c once under
 * AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle asan).  TEST INFRASTRUCTURE.
 * Shapes are chosen to hit the edges: empty and single-path runs, ragged period counts around the
 * Philox block sizes, one-bucket and many-bucket histograms, tables of one entry and above the dense
 * limit, trajectories that overflow to inf, ranks at both ends. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../smmc_oracle.c"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); fails++; } } while (0)

int main(void) {
  enum { T = 1127, TBIG = 5000 };
  float *table = malloc(sizeof(float) * TBIG);
  for (int i = 0; i < TBIG; i++) table[i] = (float)(((i * 2654435761u) >> 8) % 2000) / 100.0f - 9.5f;

  EXPECT(orc_update_fund(1000.0f, 0.5f) == 1005.0f);
  float tot[5] = {1000.0f, 0, 0, 0, 0}, rets[4] = {1.0f, -2.0f, 3.5f, -100.0f};
  orc_many_updates(rets, tot, 4);
  EXPECT(tot[4] == 0.0f && tot[1] == 1010.0f);
  orc_many_updates(rets, tot, 0);

  uint32_t raw[4], idx[64];
  orc_mt19937_raw(5489u, 4, raw);
  EXPECT(raw[0] == 3499211612u);
  const uint32_t ranges[] = {1u, 2u, 3u, T, 65536u, 2147483647u};
  for (unsigned r = 0; r < 6; r++) {
    orc_mt19937_indices(0xffffffffu, ranges[r], 64, idx);
    for (int i = 0; i < 64; i++) EXPECT(idx[i] < ranges[r]);
  }
  const uint32_t ctr[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, key[2] = {0xffffffffu, 0xffffffffu};
  uint32_t ph[4];
  orc_philox4x32_10(ctr, key, ph);
  EXPECT(ph[0] == 0x408f276du && ph[1] == 0x41c83b0eu && ph[2] == 0xa20bc7c6u && ph[3] == 0x6d5451fdu);  /* Random123 kat_vectors */

  /* Box-Muller over the corners of the uniform range */
  const uint32_t corners[] = {0u, 1u, 0x7fffffffu, 0x80000000u, 0x80000001u, 0xffffffffu, 0x00800000u, 0xff7fffffu};
  for (unsigned a = 0; a < 8; a++)
    for (unsigned b = 0; b < 8; b++) {
      float zc, zs;
      orc_box_muller(corners[a], corners[b], &zc, &zs);
      EXPECT(isfinite(zc) && isfinite(zs) && fabsf(zc) < 7.0f && fabsf(zs) < 7.0f);
      orc_box_muller_scaled(corners[a], corners[b], 0.83333f, 0.5f, &zc, &zs);
      EXPECT(isfinite(zc) && isfinite(zs));
      EXPECT(orc_bm_radius(corners[a]) >= 0.0f);
      orc_box_muller3(corners[a], corners[b], &zc, &zs);
      EXPECT(isfinite(zc) && isfinite(zs) && fabsf(zc) < 7.0f && fabsf(zs) < 7.0f);
      orc_box_muller3_scaled(corners[a], corners[b] ^ 0x00400000u, 0.83333f, 100.5f, &zc, &zs);
      EXPECT(isfinite(zc) && isfinite(zs));
      EXPECT(orc_bm3_radius(corners[a]) >= 0.0f);
    }
  EXPECT(orc_bm3_radius_scan(0, 1ull << 32, 1000003) < 1e-6);
  EXPECT(orc_bm_radius_scan(0, 1ull << 32, 1000003) < 1e-6);

  /* engine (R) */
  float *fin = malloc(sizeof(float) * 4099);
  EXPECT(orc_ref_mc_simulations(2501, 37, 1000.0f, table, T, 12345u, fin, 2) == 2);
  EXPECT(orc_ref_mc_simulations(0, 37, 1000.0f, table, T, 1u, fin, 1) == 1);
  EXPECT(orc_ref_mc_simulations(3, 0, 1000.0f, table, 1, 1u, fin, 1) == 1 && fin[2] == 1000.0f);

  /* engine (C): both modes, dense and sparse table schedules, ragged period counts */
  const uint32_t periods[] = {0, 1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 360};
  uint64_t *hist = malloc(sizeof(uint64_t) * 4096);
  float *traj = malloc(sizeof(float) * 300 * 361);
  for (int mode = 0; mode < 2; mode++)
    for (unsigned tl = 0; tl < 3; tl++)
      for (unsigned pi = 0; pi < 12; pi++) {
        orc_params p;
        memset(&p, 0, sizeof p);
        p.mode = mode;
        p.n_periods = periods[pi];
        p.seed = 0xfedcba9876543210ull;
        p.first_path = tl == 2 ? 0xfffffffffffffff0ull : (1ull << 32) - 7;  /* wraps 32 and 64 bits */
        p.n_paths = 300;
        p.initial_capital = 1000.0f;
        p.gauss_mean = 0.5f;
        p.gauss_std = 0.83333f;
        p.table = table;
        p.table_len = tl == 0 ? 1 : tl == 1 ? T : TBIG;
        p.n_bins = tl == 0 ? 1 : 4096;
        p.hist_lo = 0.0f;
        p.hist_hi = 20000.0f;
        p.below_threshold = 1000.0f;
        p.stream = (pi & 1) ? 2u : 3u;  /* both Gaussian draws (table mode ignores it) */
        orc_stats st;
        EXPECT(orc_counter_mc(&p, fin, hist, &st, traj, 2) == 0);
        EXPECT(st.count == 300);
        uint64_t total = st.underflow + st.overflow;
        for (uint32_t b = 0; b < p.n_bins; b++) total += hist[b];
        EXPECT(total == 300);
        for (int i = 0; i < 300; i++) EXPECT(traj[(size_t)i * (p.n_periods + 1) + p.n_periods] == fin[i]);
        float rr[360];
        uint32_t ii[360];
        orc_counter_path_returns(&p, p.first_path + 5, rr);
        if (mode == 0) {
          orc_counter_path_indices(&p, p.first_path + 5, ii);
          for (uint32_t k = 0; k < p.n_periods; k++) EXPECT(ii[k] < p.table_len && rr[k] == table[ii[k]]);
        }
        EXPECT(orc_counter_mc(&p, 0, 0, &st, 0, 1) == 0);  /* every output optional */
        p.n_paths = 0;
        EXPECT(orc_counter_mc(&p, fin, hist, &st, 0, 1) == 0 && st.count == 0);
      }
  {
    orc_params p;  /* table mode without a table is an error, not a crash */
    memset(&p, 0, sizeof p);
    p.n_paths = 1;
    EXPECT(orc_counter_mc(&p, fin, 0, 0, 0, 1) == -1);
    /* +900 % per period for 360 periods: inf, then the histogram's overflow bucket */
    float huge = 900.0f;
    p.table = &huge;
    p.table_len = 1;
    p.n_periods = 360;
    p.initial_capital = 1000.0f;
    p.n_bins = 8;
    p.hist_hi = 1.0f;
    orc_stats st;
    EXPECT(orc_counter_mc(&p, fin, hist, &st, 0, 1) == 0 && isinf(fin[0]) && st.overflow == 1);
  }
  EXPECT(orc_hist_bucket(NAN, 0.0f, 1.0f, 4) == 4 && orc_hist_bucket(-1.0f, 0.0f, 1.0f, 4) == -1);
  EXPECT(orc_hist_bucket(0.99999994f, 0.0f, 1.0f, 4) == 3 && orc_hist_bucket(1.0f, 0.0f, 1.0f, 4) == 4);
  EXPECT(orc_draws_per_block(0, 2048) == 8 && orc_draws_per_block(0, 2049) == 4 && orc_draws_per_block(1, 5) == 4);

  /* statistics helpers */
  float v[1000], means[4], vars[4];
  for (int i = 0; i < 1000; i++) v[i] = (float)((i * 7919) % 1000);
  orc_chunk_mean_var(v, 1000, 256, means, vars);
  orc_chunk_mean_var(v, 0, 256, means, vars);
  orc_stats st;
  orc_values_stats(v, 1000, 500.0f, 10, 0.0f, 1000.0f, &st, hist);
  EXPECT(st.count == 1000 && st.below == 500 && st.min == 0.0f && st.max == 999.0f);
  orc_values_stats(v, 0, 500.0f, 0, 0.0f, 1.0f, &st, 0);
  const uint64_t ranks[5] = {0, 250, 500, 750, 999};
  float q[5];
  EXPECT(orc_order_statistics(v, 1000, ranks, 5, q) == 0 && q[0] == 0.0f && q[2] == 500.0f && q[4] == 999.0f);
  const uint64_t bad_rank = 1000;
  EXPECT(orc_order_statistics(v, 1000, &bad_rank, 1, q) != 0);
  uint32_t first_bad = 0;
  EXPECT(orc_div100_mismatches(0x3f800000u, 0x3f800000u + 100000u, &first_bad) == 0);
  EXPECT(orc_div100_mismatches(0u, 4096u, &first_bad) > 0);  /* denormal inputs: outside the proven range */

  free(table); free(fin); free(hist); free(traj);
  printf(fails ? "oracle_driver: %d FAILURES\n" : "oracle_driver: ok\n", fails);
  return fails != 0;
}
