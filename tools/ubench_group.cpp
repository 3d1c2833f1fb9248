// ubench_group.cpp -- what the two merge back-ends of smmc_group cost:
// group creation (engines; RCCL: opening librccl + ncclCommInitAll), and per call the merge step and the
// whole statistics-only call, for SMMC_MERGE_HOST and SMMC_MERGE_RCCL over the same devices.
//
// build: g++ -O2 -std=c++17 -Iinclude tools/ubench_group.cpp -o /tmp/ubench_group -Lstock_market_monte_carlo_amd -lsmmc_hip -Wl,-rpath,$PWD/stock_market_monte_carlo_amd
// usage: ubench_group [n_devices=1] [n_paths=1000000] [n_periods=36] [calls=20]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "smmc.h"

static double ms_since(const std::chrono::steady_clock::time_point &t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv) {
  const int n_dev = argc > 1 ? std::atoi(argv[1]) : 1;
  const unsigned long long n_paths = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1000000ull;
  const unsigned periods = argc > 3 ? std::atoi(argv[3]) : 36;
  const int calls = argc > 4 ? std::atoi(argv[4]) : 20;
  std::vector<int> devs(n_dev);
  for (int i = 0; i < n_dev; ++i) devs[i] = i;
  // a first engine: the HIP runtime's own start-up is not a property of either back-end
  const auto t_rt = std::chrono::steady_clock::now();
  smmc_engine *warm = nullptr;
  if (smmc_engine_create(0, SMMC_STREAM_NEW, &warm) != SMMC_OK) {
    std::fprintf(stderr, "%s\n", smmc_last_error());
    return 1;
  }
  std::printf("{\"phase\": \"runtime start-up + first engine\", \"ms\": %.3f}\n", ms_since(t_rt));
  const uint32_t bins = 100;
  std::vector<uint64_t> hist_host(bins), hist_rccl(bins);
  smmc_stats st_host{}, st_rccl{};
  for (int merge : {SMMC_MERGE_HOST, SMMC_MERGE_RCCL, SMMC_MERGE_RCCL}) {  // RCCL twice: the second group finds librccl open
    const auto t0 = std::chrono::steady_clock::now();
    smmc_group *g = nullptr;
    if (smmc_group_create(devs.data(), n_dev, merge, &g) != SMMC_OK) {
      std::fprintf(stderr, "%s\n", smmc_last_error());
      return 1;
    }
    const double create_ms = ms_since(t0);
    double engines_ms = 0, comm_ms = 0, merge_ms = 0;
    smmc_group_timings(g, &engines_ms, &comm_ms, nullptr);
    smmc_sim sim{};
    sim.struct_size = sizeof sim;
    sim.mode = SMMC_MODE_GAUSSIAN;
    sim.seed = 0x5EED5EED5EED5EEDull;
    sim.n_paths = n_paths;
    sim.n_periods = periods;
    sim.initial_capital = 1000.f;
    sim.gauss_mean = 0.5f;
    sim.gauss_std = 0.83333f;
    sim.n_bins = bins;
    sim.hist_lo = 0.f;
    sim.hist_hi = 20000.f;
    sim.below_threshold = 1000.f;
    std::vector<double> merges, totals;
    smmc_stats st{};
    std::vector<uint64_t> hist(bins);
    for (int c = 0; c < calls + 2; ++c) {
      const auto t1 = std::chrono::steady_clock::now();
      if (smmc_group_simulate(g, &sim, nullptr, nullptr, nullptr, nullptr, &st, hist.data()) != SMMC_OK) {
        std::fprintf(stderr, "%s\n", smmc_last_error());
        return 1;
      }
      const double total = ms_since(t1);
      smmc_group_timings(g, nullptr, nullptr, &merge_ms);
      if (c >= 2) {  // two warm-up calls
        merges.push_back(merge_ms);
        totals.push_back(total);
      }
    }
    std::sort(merges.begin(), merges.end());
    std::sort(totals.begin(), totals.end());
    std::printf("{\"merge\": \"%s\", \"devices\": %d, \"group_create_ms\": %.3f, \"engines_ms\": %.3f, \"comm_init_ms\": %.3f, "
                "\"merge_step_ms_median\": %.4f, \"merge_step_ms_min\": %.4f, \"merge_step_ms_max\": %.4f, "
                "\"call_ms_median\": %.4f, \"n_paths\": %llu, \"n_periods\": %u, \"calls\": %d, \"count\": %llu, \"below\": %llu}\n",
                merge == SMMC_MERGE_HOST ? "host" : "rccl", n_dev, create_ms, engines_ms, comm_ms, merges[merges.size() / 2],
                merges.front(), merges.back(), totals[totals.size() / 2], n_paths, periods, calls,
                (unsigned long long)st.count, (unsigned long long)st.below);
    if (merge == SMMC_MERGE_HOST) { st_host = st; hist_host = hist; } else { st_rccl = st; hist_rccl = hist; }
    smmc_group_destroy(g);
  }
  const bool same = st_host.count == st_rccl.count && st_host.below == st_rccl.below && st_host.sum == st_rccl.sum &&
                    st_host.sumsq == st_rccl.sumsq && st_host.min == st_rccl.min && st_host.max == st_rccl.max && hist_host == hist_rccl;
  std::printf("{\"host_and_rccl_records_identical\": %s}\n", same ? "true" : "false");
  smmc_engine_destroy(warm);
  return same ? 0 : 2;
}
