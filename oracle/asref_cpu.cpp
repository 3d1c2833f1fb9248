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
