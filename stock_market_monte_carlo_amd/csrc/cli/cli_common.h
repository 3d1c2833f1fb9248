// cli_common.h -- shared by the benchmark_mc_* programs (drop-ins for the reference's
// examples/benchmark_mc_*.cpp: same positional arguments, same final stdout lines).
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "stock_market_monte_carlo/simulations.h"

namespace cli {

// The reference reads a fixed path (examples/benchmark_mc_cpu_v2.cpp:25).  SMMC_TABLE
// overrides it; if the file is absent the bundled synthetic table is used, and said.
inline std::vector<float> load_returns() {
  const char *env = std::getenv("SMMC_TABLE");
  const std::string path = env ? env : "data/SP500_monthly_returns.csv";
  try {
    std::vector<float> t = read_historical_returns(path);
    if (!t.empty()) {
      std::printf("returns table: %s (%zu entries)\n", path.c_str(), t.size());
      return t;
    }
  } catch (const std::exception &) {
  }
  std::vector<float> t = smmc::bundled_synthetic_returns();
  std::printf("returns table: %s not readable -> bundled SYNTHETIC table (%zu entries)\n", path.c_str(), t.size());
  return t;
}

inline bool gaussian_mode() {
  const char *m = std::getenv("SMMC_MODE");
  return m && std::strcmp(m, "gaussian") == 0;
}

// fmt's {:L} with the en_US locale: thousands separators
inline std::string grouped(long v) {
  std::string s = std::to_string(v < 0 ? -v : v), out;
  for (size_t i = 0; i < s.size(); ++i) {
    if (i && (s.size() - i) % 3 == 0) out += ',';
    out += s[i];
  }
  return (v < 0 ? "-" : "") + out;
}

struct Stopwatch {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double seconds() const {
    // the reference truncates to whole milliseconds (benchmark_mc_gpu.cpp:71)
    return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() / 1000.0;
  }
};

inline void json_line(const char *program, long n, int periods, int n_gpus, double seconds, double mean, double std_,
                      long below) {
  if (!std::getenv("SMMC_JSON")) return;
  std::printf("{\"program\": \"%s\", \"n_simulations\": %ld, \"n_periods\": %d, \"n_gpus\": %d, \"seconds\": %.6f, "
              "\"paths_per_s\": %.6g, \"mean\": %.6f, \"std\": %.6f, \"count_below\": %ld}\n",
              program, n, periods, n_gpus, seconds, seconds > 0 ? n / seconds : 0.0, mean, std_, below);
}

}  // namespace cli
