// simulations.h -- C++ drop-in for the Monte-Carlo part of the reference's public
// header (include/stock_market_monte_carlo/simulations.h of
// matthijsvk/stock_market_monte_carlo), backed by libsmmc_hip.so on an MI355X.
//
// Every function below keeps the reference's name, parameter list and ownership
// rules, so the reference's own callers (examples/benchmark_mc_*.cpp,
// examples/visualize_returns_*.cpp) compile against this header unchanged.  The
// reference location of each one is cited.  What is NOT here: the CSV-dumping demos
// (one_simulation_*, monte_carlo_*), which are file I/O, not the hot path.
//
// Self-contained on purpose: the reference header relies on fmt's transitive
// includes for <atomic> and <string>.
#ifndef SMMC_DROPIN_SIMULATIONS_H
#define SMMC_DROPIN_SIMULATIONS_H

// <chrono>, <iostream>, <random>, <vector>: the reference header includes them
// (simulations.h:1-4) and its callers rely on that; <atomic>, <string> it gets from fmt.
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstdint>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "helpers.h"  // the reference header includes it too (simulations.h:7)

// ---- compounding core (src/simulations.cpp:14-39) ------------------------------------
float update_fund(float fund_value, float period_return);                             // :14-16
void __many_updates(float *returns, float *totals, unsigned int n_periods);           // :18-22
// :24-39 -- returns n_periods + 1 values, [0] == fund_value
std::vector<float> many_updates(float fund_value, std::vector<float> &returns, unsigned int n_periods);
// the overload the reference header declares (simulations.h:11-13) but never defines
std::vector<float> many_updates(float fund_value, std::vector<float> &returns, long n_updates);

// ---- returns sources (src/simulations.cpp:41-55, 83-112) ------------------------------
std::vector<float> sample_returns_gaussian(unsigned int n, float return_mean, float return_std);   // :41-55
std::vector<float> read_historical_returns(std::string csv_fpath);                                  // :83-93
std::vector<float> sample_returns_historical(unsigned int n, std::vector<float> &historical_returns);  // :95-112

// ---- "CPU" engines of the reference, here executed on the GPU -------------------------
// src/simulations.cpp:204-266.  final_values must already hold max_n_simulations
// entries (the reference writes by index, :252); n_simulations advances while it runs.
void mc_simulations(std::atomic<long> &n_simulations, long max_n_simulations, unsigned int n_periods,
                    float initial_capital, std::vector<float> &historical_returns,
                    std::vector<float> &final_values);
// src/simulations.cpp:139-202.  mc_data and final_values must hold max_n_simulations
// entries; mc_data[i] becomes the n_periods + 1 values of path i.
void mc_simulations_keepdata(std::atomic<long> &n_simulations, long max_n_simulations, unsigned int n_periods,
                             float initial_capital, std::vector<float> &historical_returns,
                             std::vector<std::vector<float>> &mc_data, std::vector<float> &final_values);

// ---- GPU engines (src/simulations.cu:661-697) ------------------------------------------
// totals is replaced by a vector of max_n_simulations final values (:643-644).
void mc_simulations_gpu(std::atomic<long> &n_simulations, long max_n_simulations, int n_periods,
                        float initial_capital, std::vector<float> &returns, std::vector<float> &totals,
                        int n_gpus);
// means / variances are resized to ceil(max_n_simulations / 256) (:429-432); throws
// std::invalid_argument unless n_gpus == 1 (:693).
void mc_simulations_gpu_reduceBlock(std::atomic<long> &n_simulations, long max_n_simulations, int n_periods,
                                    float initial_capital, std::vector<float> &returns,
                                    std::vector<float> &means, std::vector<float> &variances, int n_gpus);

// ---- mean of a host array on the GPU (src/simulations.cu:269-341) ----------------------
// Mean of the first n entries of vec; double accumulation on the device (the reference
// sums in float through an in-place strided tree and indexes with int).
float reduce_mean_gpu(std::vector<float> &vec, long n);

// ---- additions (not in the reference) ---------------------------------------------------
namespace smmc {

// The reference has no seed (std::random_device per path).  The drop-in draws one
// 64-bit seed per call from std::random_device unless a seed is fixed here or through
// the environment variable SMMC_SEED.  fix_seed(false, ...) returns to random seeds.
void fix_seed(bool fixed, std::uint64_t seed);

// v becomes n zeros (previous contents discarded), with a new allocation's pages mapped beforehand (two threads of
// madvise(MADV_POPULATE_WRITE)): 17 ms instead of 60 ms for the 400 MB result of 1e8 paths on the GPU
// box's host.  What mc_simulations_gpu does for its callee-sized `totals`; callers that pre-size the
// vector of mc_simulations (examples/benchmark_mc_cpu_v2.cpp:26) can use it too.  Contents: zeros.
void resize_prefaulted(std::vector<float> &v, std::size_t n);

// Gaussian-returns variant of mc_simulations_gpu (mean / std in percent per period,
// examples/monte_carlo_simulated.cpp:11-12): the mode BASELINE configs 2, 4 and 5 name.
void mc_simulations_gpu_gaussian(std::atomic<long> &n_simulations, long max_n_simulations, int n_periods,
                                 float initial_capital, float return_mean, float return_std,
                                 std::vector<float> &totals, int n_gpus);

// Fused statistics of a run, replacing the host passes of examples/benchmark_mc_gpu.cpp:7-41.
struct Summary {
  std::uint64_t count = 0, below = 0, underflow = 0, overflow = 0;
  double sum = 0, sumsq = 0;
  float min = 0, max = 0;
  std::vector<std::uint64_t> hist;
  double mean() const { return count ? sum / double(count) : 0.0; }
  double stddev() const;  // population
};
// Runs the simulation WITHOUT materialising final values on the host: only the
// statistics record crosses PCIe.  gaussian == false draws from `returns`.
Summary mc_summary(long max_n_simulations, int n_periods, float initial_capital, bool gaussian,
                   std::vector<float> &returns, float return_mean, float return_std, float below_threshold,
                   unsigned n_bins, float hist_lo, float hist_hi, int n_gpus);

// The statistics helpers the reference keeps in its example programs
// (examples/visualize_returns_cpu_v2.cpp:83-138), executed on the GPU: the prefix
// vec[0 .. n_el) is copied to the device once per call.  quartiles becomes
// {min, Q1, Q2, Q3, max} by exact selection (no sort, the input is not modified).
void update_quartiles(std::vector<float> &quartiles, std::vector<float> &vec, long n_el);
void update_mean_std(float &mean, float &std, std::vector<float> &v, long n_el);
long update_count_below_min(float &min_final_amount, const std::vector<float> &final_values, long n_simulations);

// The 1127-entry synthetic returns table bundled with the library (percent units),
// used by the CLIs when data/SP500_monthly_returns.csv is absent.
std::vector<float> bundled_synthetic_returns();

}  // namespace smmc

#endif
