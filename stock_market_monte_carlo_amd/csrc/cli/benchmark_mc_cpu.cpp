// benchmark_mc_cpu <n_months> <n_simulations>
// Drop-in for the reference's examples/benchmark_mc_cpu.cpp: keeps every trajectory
// (mc_simulations_keepdata); the caller pre-sizes mc_data and final_values (:26-28).
#include "cli_common.h"

int main(int argc, char **argv) {
  std::printf("argc: %d\n", argc);
  if (argc != 3) {
    std::printf("usage: visualize_returns <n_months> <n_simulations>, eg visualize_returns 360 100000");
    return 0;
  }
  const unsigned int n_periods = static_cast<unsigned int>(std::strtol(argv[1], nullptr, 10));
  const long max_n = std::strtol(argv[2], nullptr, 10);
  std::printf("n_periods: %u | max_n_simulations: %ld\n", n_periods, max_n);

  const float initial_capital = 1000;
  std::vector<float> returns = cli::load_returns();
  std::vector<std::vector<float>> mc_data(static_cast<size_t>(max_n), std::vector<float>(n_periods));
  std::vector<float> final_values(static_cast<size_t>(max_n), initial_capital);
  std::atomic<long> n_simulations{0};
  try {
    cli::Stopwatch sw;
    mc_simulations_keepdata(n_simulations, max_n, n_periods, initial_capital, returns, mc_data, final_values);
    const double secs = sw.seconds();
    std::printf("All %ld simulation done in %g s!\n", n_simulations.load(), secs);
    cli::json_line("benchmark_mc_cpu", max_n, int(n_periods), 1, secs, 0, 0, 0);
  } catch (const std::exception &ex) {
    std::fprintf(stderr, "benchmark_mc_cpu: %s\n", ex.what());
    return 1;
  }
  return 0;
}
