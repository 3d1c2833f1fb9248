// benchmark_mc_gpu_reduceBlock <n_gpus> <n_months> <n_simulations>
// Drop-in for the reference's examples/benchmark_mc_gpu_reduceBlock.cpp: per-256-path
// means and variances come back from the device; the host merges them the way the
// reference does (:7-26: mean of means, sqrt of the mean variance) and estimates the
// count below the initial capital with the normal CDF (:28-31, :65-70).  A second block
// prints the EXACT statistics from the fused on-device reduction, which the reference
// cannot produce without copying every final value back.
#include <cmath>
#include <numeric>

#include "cli_common.h"

static double normal_cdf(double v) { return 0.5 * std::erfc(-v * M_SQRT1_2); }

int main(int argc, char **argv) {
  std::printf("argc: %d\n", argc);
  if (argc != 4) {
    std::printf("usage: benchmark_mc_gpu_reduceBlock <n_gpus> <n_months> <n_simulations>, eg "
                "benchmark_mc_gpu_reduceBlock 1 360 100000");
    return 0;
  }
  const int n_gpus = std::atoi(argv[1]);
  const int n_periods = std::atoi(argv[2]);
  const long max_n = std::strtol(argv[3], nullptr, 10);
  std::printf("n_periods: %d | max_n_simulations: %ld\n", n_periods, max_n);

  const float initial_capital = 1000;
  std::vector<float> returns = cli::load_returns();
  std::vector<float> means, variances;
  std::atomic<long> n_simulations{0};
  try {
    cli::Stopwatch sw;
    mc_simulations_gpu_reduceBlock(n_simulations, max_n, n_periods, initial_capital, returns, means, variances, n_gpus);
    std::printf("n_simulations: %ld\n", n_simulations.load());
    const double secs = sw.seconds();
    std::printf("All %ld simulation done in %g s!\n", n_simulations.load(), secs);

    const double sum = std::accumulate(means.begin(), means.end(), 0.0);
    const float mean = means.empty() ? 0.f : float(sum) / means.size();
    double vsum = 0;
    for (float v : variances) vsum += v;
    std::printf("Sum var: %f\n", vsum);
    const float sd = variances.empty() ? 0.f : std::sqrt(float(vsum / variances.size()));
    std::printf("mean: %.2f | std: %.2f \n", mean, sd);
    const float prob = float(normal_cdf((initial_capital - mean) / sd));
    const long below = long(max_n * prob);
    std::printf("count_below %.1f: %s (%4f%%) \n", initial_capital, cli::grouped(below).c_str(),
                max_n ? 100 * float(below) / max_n : 0.f);
    std::printf("prob below min: %.3f%% \n", 100 * prob);

    // exact, from the fused reduction (new; same seed policy, fresh run)
    cli::Stopwatch sw2;
    smmc::Summary s = smmc::mc_summary(max_n, n_periods, initial_capital, cli::gaussian_mode(), returns, 0.5f, 0.83333f,
                                       initial_capital, 100, 0.f, 20000.f, n_gpus);
    std::printf("exact (fused on-device reduction, %g s): mean: %.2f | std: %.2f | count_below %.1f: %s (%4f%%) | min %.2f max %.2f\n",
                sw2.seconds(), s.mean(), s.stddev(), initial_capital, cli::grouped(long(s.below)).c_str(),
                s.count ? 100.0 * double(s.below) / double(s.count) : 0.0, s.min, s.max);
    cli::json_line("benchmark_mc_gpu_reduceBlock", max_n, n_periods, n_gpus, secs, mean, sd, below);
  } catch (const std::exception &ex) {
    std::fprintf(stderr, "benchmark_mc_gpu_reduceBlock: %s\n", ex.what());
    return 1;
  }
  return 0;
}
