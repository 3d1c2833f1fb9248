// asref_cpu.cpp -- CPU baseline variant (i) of SURVEY section 8d (TEST INFRASTRUCTURE: only
// bench.py's cpu_baseline leg and tests/ may load this).
//
// oracle/smmc_oracle.c's engine (R) seeds path `id` with seed0 + id so that its results are
// reproducible.  The reference does not: every path constructs a std::random_device, reads it
// once and seeds a fresh std::mt19937 with it (src/simulations.cpp:245-247), which is where most
// of its CPU time goes.  This file restates that loop with the real libstdc++
// <random> classes so the cost of the reference's own seeding can be timed on the GPU box's host:
//
//   src/simulations.cpp:213-231  blocks of 1000 paths, OpenMP schedule(dynamic), hw - 1 threads
//   src/simulations.cpp:240-252  per path: random_device -> mt19937 -> uniform_int_distribution,
//                                n_periods x update_fund, final_values[id] = total
//   src/simulations.cpp:14-16    update_fund = fund * (100.0f + r) / 100
//
// Results are NOT reproducible (by construction); tests only check distribution-level agreement
// with engine (R).  With a non-zero `fixed_seed0` the random_device is replaced by seed0 + id, and the
// function must then equal engine (R) bit for bit -- that is how the restatement itself is checked.
#include <cstdint>
#include <random>

#ifdef _OPENMP
#include <omp.h>
#endif

extern "C" __attribute__((visibility("default")))
int orc_asref_mc_simulations(int64_t n_paths, uint32_t n_periods, float initial_capital, const float *table,
                             uint32_t table_len, float *final_values, int n_threads, int use_fixed_seed,
                             uint32_t fixed_seed0) {
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
    const int64_t first = b * block_size;
    const int64_t last = first + block_size < n_paths ? first + block_size : n_paths;
    for (int64_t id = first; id < last; id++) {
      uint32_t seed;
      if (use_fixed_seed) {
        seed = static_cast<uint32_t>(fixed_seed0 + static_cast<uint64_t>(id));
      } else {
        std::random_device rd;
        seed = rd();
      }
      std::mt19937 rng(seed);
      std::uniform_int_distribution<int> uni(0, static_cast<int>(table_len) - 1);
      float total = initial_capital;
      for (uint32_t i = 0; i < n_periods; i++) {
        const float a = 100.0f + table[uni(rng)];
        const float m = total * a;
        total = m / 100.0f;
      }
      final_values[id] = total;
    }
  }
  return used;
}

// BASELINE configs[0] as written: "Gaussian returns, single-thread CPU reference (fixed seed)".  The
// reference's Gaussian path is its CSV demo: one_simulation_gaussian (src/simulations.cpp:57-67) draws
// n_periods returns with sample_returns_gaussian (:41-55: std::default_random_engine seeded from the
// clock, std::normal_distribution<float>(mean, std) in percent, examples/monte_carlo_simulated.cpp:11-12)
// and compounds them with many_updates (:24-39).  Restated here with the same libstdc++ classes, the
// seed of path id fixed to a hash of seed0 + id (the reference's is the clock; default_random_engine is the
// minstd_rand0 LCG, whose first outputs are linear in the seed, so neighbouring seeds as they are would
// correlate neighbouring paths), the draw and the update fused
// (no returns / values vectors, no CSV file), and mc_simulations' OpenMP block structure (:213-231) so
// that it can also be timed on all cores.  Timed by bench.py's cpu_baseline leg next to the GPU's
// Gaussian headline; deterministic for a given seed0, independent of the thread count.
extern "C" __attribute__((visibility("default")))
int orc_asref_gaussian_mc(int64_t n_paths, uint32_t n_periods, float initial_capital, float return_mean,
                          float return_std, uint32_t seed0, float *final_values, int n_threads) {
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
    const int64_t first = b * block_size;
    const int64_t last = first + block_size < n_paths ? first + block_size : n_paths;
    for (int64_t id = first; id < last; id++) {
      uint32_t h = static_cast<uint32_t>(seed0 + static_cast<uint64_t>(id));  // murmur3 finalizer
      h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
      std::default_random_engine e(h);
      std::normal_distribution<float> distN(return_mean, return_std);
      float total = initial_capital;
      for (uint32_t i = 0; i < n_periods; i++) {
        const float a = 100.0f + distN(e);  // update_fund, src/simulations.cpp:14-16
        const float m = total * a;
        total = m / 100.0f;
      }
      final_values[id] = total;
    }
  }
  return used;
}
