// ubench_startup.cpp -- where a cold `benchmark_mc_gpu 1 360 100000000` spends its wall time:
// C-ABI calls timed one by one.  g++ -O2 -Iinclude tools/ubench_startup.cpp -Lstock_market_monte_carlo_amd -lsmmc_hip
#include <chrono>
#include <cstdio>
#include <vector>
#include "smmc.h"
#include "stock_market_monte_carlo/simulations.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  double t0 = now();
  int n = 0;
  smmc_device_count(&n);
  std::printf("smmc_device_count (HIP start-up)   %7.1f ms (%d devices)\n", (now() - t0) * 1e3, n);
  t0 = now();
  smmc_engine *e = nullptr;
  if (smmc_engine_create(0, SMMC_STREAM_NEW, &e)) { std::printf("create failed: %s\n", smmc_last_error()); return 1; }
  std::printf("smmc_engine_create                 %7.1f ms\n", (now() - t0) * 1e3);
  std::vector<float> table = smmc::bundled_synthetic_returns();
  t0 = now();
  smmc_engine_set_table(e, table.data(), (uint32_t)table.size());
  std::printf("smmc_engine_set_table              %7.1f ms\n", (now() - t0) * 1e3);
  const size_t N = 100000000;
  t0 = now();
  std::vector<float> out;
  smmc::resize_prefaulted(out, N);
  std::printf("resize_prefaulted(1e8)             %7.1f ms\n", (now() - t0) * 1e3);
  smmc_sim sim{};
  sim.struct_size = sizeof sim; sim.mode = SMMC_MODE_TABLE; sim.seed = 1; sim.n_paths = N; sim.n_periods = 360;
  sim.initial_capital = 1000.f; sim.below_threshold = 1000.f;
  for (int rep = 0; rep < 3; ++rep) {
    t0 = now();
    int rc = smmc_engine_simulate_to_host(e, &sim, out.data(), nullptr, nullptr, nullptr, nullptr, nullptr);
    std::printf("smmc_engine_simulate_to_host #%d    %7.1f ms (rc %d)\n", rep, (now() - t0) * 1e3, rc);
  }
  t0 = now();
  smmc_engine_destroy(e);
  std::printf("smmc_engine_destroy                %7.1f ms\n", (now() - t0) * 1e3);
  return 0;
}
