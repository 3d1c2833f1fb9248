// benchmark_mc_gpu <n_gpus> <n_months> <n_simulations>
// Drop-in for the reference's examples/benchmark_mc_gpu.cpp: runs mc_simulations_gpu,
// then the host-side mean/std (double accumulators, :7-28) and count-below (:30-41)
// passes over the returned final values, and prints the same closing lines (:72-80).
// SMMC_MODE=gaussian switches the draw to N(0.5 %, 0.83333 %) (BASELINE configs 2/4/5).
#include <cmath>

#include "cli_common.h"

int main(int argc, char **argv) {
  std::printf("expected arguments: <n_gpus>, <n_periods> <n_simulations>");
  std::printf("argc: %d\n", argc);
  if (argc != 4) {
    std::printf("usage: benchmark_mc_gpu <n_gpus> <n_months> <n_simulations>, eg benchmark_mc_gpu 1 360 100000");
    return 0;  // the reference exits 0 on a usage error (:60)
  }
  const int n_gpus = std::atoi(argv[1]);
  const int n_periods = std::atoi(argv[2]);
  const long max_n = std::strtol(argv[3], nullptr, 10);
  std::printf("n_periods: %d | max_n_simulations: %ld\n", n_periods, max_n);

  const float initial_capital = 1000;
  std::vector<float> returns = cli::load_returns();
  std::vector<float> final_values;
  std::atomic<long> n_simulations{0};
  try {
    cli::Stopwatch sw;
    if (cli::gaussian_mode())
      smmc::mc_simulations_gpu_gaussian(n_simulations, max_n, n_periods, initial_capital, 0.5f, 0.83333f, final_values, n_gpus);
    else
      mc_simulations_gpu(n_simulations, max_n, n_periods, initial_capital, returns, final_values, n_gpus);
    const double secs = sw.seconds();
    std::printf("All %ld simulation done in %g s!\n", n_simulations.load(), secs);

    double sum = 0;
    for (float v : final_values) sum += v;
    const float mean = final_values.empty() ? 0.f : float(sum / final_values.size());
    sum = 0;
    for (float v : final_values) {
      const float d = v - mean;
      sum += d * d;
    }
    const float sd = final_values.empty() ? 0.f : std::sqrt(float(sum / final_values.size()));
    std::printf("mean: %.2f | std: %.2f \n", mean, sd);
    long below = 0;
    for (long i = 0; i < max_n; ++i)
      if (final_values[i] < initial_capital) ++below;
    std::printf("count_below %.1f: %s (%4f%%)\n", initial_capital, cli::grouped(below).c_str(),
                max_n ? 100 * float(below) / max_n : 0.f);
    cli::json_line("benchmark_mc_gpu", max_n, n_periods, n_gpus, secs, mean, sd, below);
  } catch (const std::exception &ex) {
    std::fprintf(stderr, "benchmark_mc_gpu: %s\n", ex.what());
    return 1;
  }
  return 0;
}
