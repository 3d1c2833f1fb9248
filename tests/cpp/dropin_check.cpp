// dropin_check.cpp -- exercises the reference-named C++ API (simulations.h) the way the
// reference's callers do and prints one JSON object for tests/test_dropin_gpu.py.
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <cstring>
#include <stdexcept>
#include <thread>

#include "stock_market_monte_carlo/simulations.h"

static std::uint64_t fnv(const std::vector<float> &v) {
  std::uint64_t h = 0xCBF29CE484222325ull;
  for (float f : v) {
    unsigned char b[4];
    std::memcpy(b, &f, 4);
    for (unsigned char c : b) h = (h ^ c) * 0x100000001B3ull;
  }
  return h;
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 20000;
  const int periods = argc > 2 ? std::atoi(argv[2]) : 36;
  std::vector<float> table = read_historical_returns("data/SP500_monthly_returns.csv");
  smmc::fix_seed(true, 4242);
  std::atomic<long> counter{0};

  std::vector<float> gpu_totals;  // callee sizes it
  mc_simulations_gpu(counter, n, periods, 1000.f, table, gpu_totals, 1);
  const long counter_gpu = counter;

  std::vector<float> cpu_final(n, 1000.f);  // caller sizes it
  long seen_mid = -1;
  {
    // the reference's GUI polls n_simulations from another thread while the engine runs
    std::atomic<bool> done{false};
    std::thread poll([&] { while (!done) { long c = counter; if (c > 0 && c < n) seen_mid = c; std::this_thread::yield(); } });
    mc_simulations(counter, n, static_cast<unsigned>(periods), 1000.f, table, cpu_final);
    done = true;
    poll.join();
  }

  std::vector<float> means, variances;
  mc_simulations_gpu_reduceBlock(counter, n, periods, 1000.f, table, means, variances, 1);
  bool threw = false;
  try {
    mc_simulations_gpu_reduceBlock(counter, n, periods, 1000.f, table, means, variances, 2);
  } catch (const std::invalid_argument &) {
    threw = true;
  }

  const long nk = n < 3000 ? n : 3000;
  std::vector<std::vector<float>> mc_data(nk);
  std::vector<float> keep_final(nk, -1.f);
  mc_simulations_keepdata(counter, nk, static_cast<unsigned>(periods), 1000.f, table, mc_data, keep_final);
  bool rows_ok = true;
  for (long i = 0; i < nk; ++i)
    rows_ok = rows_ok && mc_data[i].size() == size_t(periods) + 1 && mc_data[i][0] == 1000.f && mc_data[i].back() == keep_final[i];
  // a trajectory is many_updates of something: re-run the recurrence from its own ratios is not possible,
  // so check the chain with update_fund on the implied returns instead (values[i+1] = update_fund(values[i], r))
  std::vector<float> rets = {1.f, -2.f, 3.5f};
  std::vector<float> mu = many_updates(1000.f, rets, 3u);
  std::vector<float> mu_long = many_updates(1000.f, rets, 3l);

  std::vector<float> gauss;
  smmc::mc_simulations_gpu_gaussian(counter, n, periods, 1000.f, 0.5f, 0.83333f, gauss, 1);
  if (const char *path = std::getenv("SMMC_DROPIN_DUMP_GAUSS")) {  // raw binary32 final values, for tests/test_gaussian_reference_gpu.py
    if (FILE *f = std::fopen(path, "wb")) {
      std::fwrite(gauss.data(), sizeof(float), gauss.size(), f);
      std::fclose(f);
    }
  }
  smmc::Summary s = smmc::mc_summary(n, periods, 1000.f, true, table, 0.5f, 0.83333f, 1000.f, 50, 0.f, 5000.f, 1);
  std::uint64_t hist_total = 0;
  for (auto c : s.hist) hist_total += c;

  // statistics helpers of the reference's example programs, on the host vector of final values
  std::vector<float> quart;
  smmc::update_quartiles(quart, gpu_totals, n);
  float hmean = 0, hstd = 0, thr = 1200.f;
  smmc::update_mean_std(hmean, hstd, gpu_totals, n);
  const long hbelow = smmc::update_count_below_min(thr, gpu_totals, n);
  std::vector<float> ramp(1000003);
  for (size_t i = 0; i < ramp.size(); ++i) ramp[i] = float(i);
  const float ramp_mean = reduce_mean_gpu(ramp, long(ramp.size()));
  // CSV writers (src/helpers.cpp)
  std::vector<float> wr = {1.5f, -2.25f}, wv = {1000.f, 1015.f, 992.1625f};
  write_data_file("dropin_check.csv", wr, wv);
  write_vector_file("outputs/dropin_check_vec.csv", wv);

  // two engines at once from two host threads, as the reference's GUI starts them
  // (examples/visualize_returns_cpu_v2.cpp:185-202), while this thread polls a counter
  std::vector<float> conc_final(n, 1000.f), conc_keep_final(nk, -1.f);
  std::vector<std::vector<float>> conc_data(nk);
  std::atomic<long> ca{0}, cb{0};
  {
    std::thread ta([&] { mc_simulations(ca, n, static_cast<unsigned>(periods), 1000.f, table, conc_final); });
    std::thread tb([&] { mc_simulations_keepdata(cb, nk, static_cast<unsigned>(periods), 1000.f, table, conc_data, conc_keep_final); });
    while (ca < n || cb < nk) std::this_thread::yield();
    ta.join();
    tb.join();
  }
  const bool concurrent_ok = fnv(conc_final) == fnv(cpu_final) && fnv(conc_keep_final) == fnv(keep_final) &&
                             conc_data[nk - 1] == mc_data[nk - 1];

  // n_gpus > 1 (SMMC_DEVICE_MAP lets three shards share this box's one GPU): the path ids are global,
  // so the result must not depend on the split; the N mod 3 remainder is kept (src/simulations.cu:602-603 drops it)
  bool multi_ran = false, multi_same = false, multi_summary_same = false;
  long multi_counter = -1;
  const long n_multi = (n + 1) % 3 ? n + 1 : n + 2;  // never divisible by 3
  if (std::getenv("SMMC_DEVICE_MAP")) {
    multi_ran = true;
    std::vector<float> one, three;
    mc_simulations_gpu(counter, n_multi, periods, 1000.f, table, one, 1);
    mc_simulations_gpu(counter, n_multi, periods, 1000.f, table, three, 3);
    multi_counter = counter;
    multi_same = one.size() == size_t(n_multi) && one == three;
    smmc::Summary s1 = smmc::mc_summary(n_multi, periods, 1000.f, false, table, 0.f, 0.f, 1000.f, 64, 0.f, 8000.f, 1);
    smmc::Summary s3 = smmc::mc_summary(n_multi, periods, 1000.f, false, table, 0.f, 0.f, 1000.f, 64, 0.f, 8000.f, 3);
    multi_summary_same = s1.count == std::uint64_t(n_multi) && s3.count == s1.count && s3.below == s1.below &&
                         s3.hist == s1.hist && s3.underflow == s1.underflow && s3.overflow == s1.overflow &&
                         s3.min == s1.min && s3.max == s1.max && std::fabs(s3.sum - s1.sum) <= 1e-12 * std::fabs(s1.sum) &&
                         std::fabs(s3.sumsq - s1.sumsq) <= 1e-12 * std::fabs(s1.sumsq);
  }

  // progress granularity: a long run must be seen advancing in several steps by a polling thread
  // (the reference advances n_simulations every 1000 paths, src/simulations.cpp:254)
  size_t progress_steps = 0;
  bool progress_monotone = true;
  {
    const long n_big = 24000000;
    std::vector<float> big(n_big);
    std::atomic<long> cbig{0};
    std::atomic<bool> done{false};
    std::set<long> seen;
    std::thread poll([&] {
      long prev = 0;
      while (!done) {
        const long c = cbig;
        if (c < prev) progress_monotone = false;
        prev = c;
        if (c > 0 && c < n_big) seen.insert(c);
        std::this_thread::yield();
      }
    });
    mc_simulations(cbig, n_big, 8u, 1000.f, table, big);
    done = true;
    poll.join();
    progress_steps = seen.size();
    progress_monotone = progress_monotone && cbig == n_big;
  }

  // update_mean_std on a large offset with a small spread (the one-pass variance must be formed
  // from the double mean), and on an all-equal vector (std exactly 0, never NaN)
  float lv_mean = 0, lv_std = 0, eq_mean = 0, eq_std = -1;
  {
    std::vector<float> lv(1000000), eq(100000, 1000.013f);
    for (size_t i = 0; i < lv.size(); ++i) lv[i] = 1000.013f + 0.05f * std::sin(0.001f * float(i));
    smmc::update_mean_std(lv_mean, lv_std, lv, long(lv.size()));
    smmc::update_mean_std(eq_mean, eq_std, eq, long(eq.size()));
  }

  std::printf("{\"concurrent_ok\": %s, ", concurrent_ok ? "true" : "false");
  std::printf("\"multi_ran\": %s, \"multi_same\": %s, \"multi_summary_same\": %s, \"multi_counter\": %ld, \"n_multi\": %ld, ",
              multi_ran ? "true" : "false", multi_same ? "true" : "false", multi_summary_same ? "true" : "false",
              multi_counter, n_multi);
  std::printf("\"progress_steps\": %zu, \"progress_monotone\": %s, \"lv_mean\": %.9g, \"lv_std\": %.9g, "
              "\"eq_mean\": %.9g, \"eq_std\": %.9g, ",
              progress_steps, progress_monotone ? "true" : "false", lv_mean, lv_std, eq_mean, eq_std);
  std::printf("\"quart\": [%.9g, %.9g, %.9g, %.9g, %.9g], \"hmean\": %.9g, \"hstd\": %.9g, \"hbelow\": %ld, \"ramp_mean\": %.9g, ",
              quart[0], quart[1], quart[2], quart[3], quart[4], hmean, hstd, hbelow, ramp_mean);
  std::printf("\"n\": %ld, \"gpu_hash\": %" PRIu64 ", \"cpu_hash\": %" PRIu64 ", \"counter_gpu\": %ld, \"seen_mid\": %ld, "
              "\"n_means\": %zu, \"mean0\": %.9g, \"var0\": %.9g, \"threw\": %s, \"rows_ok\": %s, \"keep_hash\": %" PRIu64 ", "
              "\"mu\": [%.9g, %.9g, %.9g, %.9g], \"mu_long_same\": %s, \"update_fund\": %.9g, "
              "\"gauss_hash\": %" PRIu64 ", \"sum_count\": %" PRIu64 ", \"sum_mean\": %.17g, \"sum_below\": %" PRIu64 ", \"hist_total\": %" PRIu64 ", "
              "\"bundled\": %zu, \"sample_hist\": %zu, \"sample_gauss\": %zu}\n",
              n, fnv(gpu_totals), fnv(cpu_final), counter_gpu, seen_mid, means.size(), means[0], variances[0],
              threw ? "true" : "false", rows_ok ? "true" : "false", fnv(keep_final), mu[0], mu[1], mu[2], mu[3],
              mu == mu_long ? "true" : "false", update_fund(1000.f, 0.5f), fnv(gauss), s.count, s.mean(), s.below,
              hist_total + s.underflow + s.overflow, smmc::bundled_synthetic_returns().size(),
              sample_returns_historical(17, table).size(), sample_returns_gaussian(9, 0.5f, 0.8f).size());
  return 0;
}
