// benchmark_reduce_mean <n>   (the reference's target name is reduce_mean / compute_avg)
// Drop-in for the reference's examples/benchmark_reduce_mean.cpp: mean of 0..n-1 on the
// CPU (double accumulate, :30-31) and through reduce_mean_gpu, printed side by side (:43).
#include <numeric>

#include "cli_common.h"

int main(int argc, char **argv) {
  std::printf("argc: %d\n", argc);
  if (argc != 2) {
    std::printf("usage: compute_avg <n>");
    return 0;
  }
  const long n = std::strtol(argv[1], nullptr, 10);
  std::printf("n: %ld\n", n);
  cli::Stopwatch setup;
  std::vector<float> vec(static_cast<size_t>(n));
  for (long i = 0; i < n; ++i) vec[i] = float(i);  // long index: the reference's int overflows beyond 2^31
  std::printf("setup of vector took %g s!\n", setup.seconds());
  try {
    cli::Stopwatch cpu;
    const double sum = std::accumulate(vec.begin(), vec.end(), 0.0);
    const float mean_cpu = float(sum) / vec.size();
    std::printf("CPU took %g s!\n", cpu.seconds());
    cli::Stopwatch gpu;
    const float mean_gpu = reduce_mean_gpu(vec, long(vec.size()));
    std::printf("GPU took %g s!\n", gpu.seconds());
    std::printf("mean_cpu: %.2f | mean_gpu: %.2f \n", mean_cpu, mean_gpu);
  } catch (const std::exception &ex) {
    std::fprintf(stderr, "benchmark_reduce_mean: %s\n", ex.what());
    return 1;
  }
  return 0;
}
