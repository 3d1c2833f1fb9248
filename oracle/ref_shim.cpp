// ref_shim.cpp -- C entry points into the reference's OWN caller code (test infrastructure).
//
// oracle/Makefile target `_ref` compiles, from the sources where they lie under
// /root/reference and without touching them:
//   src/helpers.cpp                               print_vector / write_vector_file / write_data_file
//   examples/benchmark_mc_gpu.cpp                 update_mean_std, update_count_below_min  (+ its main)
//   examples/benchmark_mc_gpu_reduceBlock.cpp     block-merge update_mean_std, normal-CDF estimator (+ main)
//   examples/benchmark_mc_cpu_v2.cpp, benchmark_mc_cpu.cpp   (mains only)
// against include/stock_market_monte_carlo/simulations.h and libsmmc_hip.so of THIS repository
// (that is the drop-in claim) and the {fmt} 12.1 headers that ship inside the image's PyTorch
// (header-only mode; a real library present in the image, not a stand-in).  The mains are
// renamed with -Dmain=..., a compiler flag; no reference source is modified or copied.
// The reference ENGINE (src/simulations.cpp) stays unbuildable: it needs "csv.h".
//
// This file (ours) only forwards to those reference functions so that Python can call them.
#include <cstdint>
#include <string>
#include <vector>

// declarations of the reference functions as their own files define them
void update_mean_std(float &mean, float &std, std::vector<float> &v, long n_el);                     // benchmark_mc_gpu.cpp:7
long update_count_below_min(float &min_final_amount, const std::vector<float> &final_values, long n);  // benchmark_mc_gpu.cpp:30
void update_mean_std(float &mean, float &std, std::vector<float> &means, std::vector<float> &variances);  // ..._reduceBlock.cpp:7
long update_count_below_min(float &min_final_amount, float mean, float std, long n_simulations);       // ..._reduceBlock.cpp:65
void write_vector_file(const std::string fname, std::vector<float> &v);                               // src/helpers.cpp:18
void write_data_file(const std::string fname, std::vector<float> &returns, std::vector<float> &values);  // src/helpers.cpp:23
int ref_main_benchmark_mc_cpu_v2(int argc, char *argv[]);
int ref_main_benchmark_mc_cpu(int argc, char *argv[]);

extern "C" {

void ref_update_mean_std(const float *v, long n, float *mean, float *std) {
  std::vector<float> vec(v, v + n);
  update_mean_std(*mean, *std, vec, n);
}

long ref_update_count_below_min(const float *v, long n, float threshold) {
  std::vector<float> vec(v, v + n);
  return update_count_below_min(threshold, vec, n);
}

void ref_merge_block_stats(const float *means, const float *variances, long n_blocks, float *mean, float *std) {
  std::vector<float> m(means, means + n_blocks), q(variances, variances + n_blocks);
  update_mean_std(*mean, *std, m, q);
}

long ref_estimate_count_below(float threshold, float mean, float std, long n) {
  return update_count_below_min(threshold, mean, std, n);
}

void ref_write_data_file(const char *fname, const float *returns, long n_returns, const float *values, long n_values) {
  std::vector<float> r(returns, returns + n_returns), v(values, values + n_values);
  write_data_file(fname, r, v);
}

void ref_write_vector_file(const char *fname, const float *values, long n) {
  std::vector<float> v(values, values + n);
  write_vector_file(fname, v);
}

int ref_run_benchmark_mc_cpu_v2(const char *n_periods, const char *n_sims) {
  char prog[] = "benchmark_mc_cpu_v2";
  std::string a = n_periods, b = n_sims;
  char *argv[] = {prog, a.data(), b.data(), nullptr};
  return ref_main_benchmark_mc_cpu_v2(3, argv);
}

int ref_run_benchmark_mc_cpu(const char *n_periods, const char *n_sims) {
  char prog[] = "benchmark_mc_cpu";
  std::string a = n_periods, b = n_sims;
  char *argv[] = {prog, a.data(), b.data(), nullptr};
  return ref_main_benchmark_mc_cpu(3, argv);
}
}
