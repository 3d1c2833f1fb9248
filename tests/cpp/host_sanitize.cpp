// host_sanitize.cpp -- the product's host-only behaviour under ASan/UBSan and TSan (CPU build,
// make -C oracle asan tsan): scalar functions, CSV reader and writers, samplers, the statistics-record
// helpers, argument checking and the "no device" error paths of the C ABI and of the C++ drop-in layer
// -- from several threads at once, as the reference's GUI callers use the API
// (examples/visualize_returns_cpu_v2.cpp:185-202).
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>

#include "smmc.h"
#include "stock_market_monte_carlo/simulations.h"

static std::atomic<int> fails{0};
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static void scalars_and_records() {
  EXPECT(update_fund(1000.f, 0.5f) == 1005.f);
  std::vector<float> r = {1.f, -2.f, 3.5f};
  std::vector<float> t = many_updates(1000.f, r, 3u);
  EXPECT(t.size() == 4 && t[0] == 1000.f && t == many_updates(1000.f, r, 3l));
  EXPECT(many_updates(5.f, r, 0u).size() == 1);
  bool threw = false;
  try { many_updates(1000.f, r, 4u); } catch (const std::out_of_range &) { threw = true; }
  EXPECT(threw);
  float totals[4] = {1000.f, 0, 0, 0};
  __many_updates(r.data(), totals, 3);
  EXPECT(totals[3] == t[3]);
  // packed statistics records
  const uint32_t bins = 7;
  std::vector<char> a(smmc_stats_bytes(bins), 0), b(smmc_stats_bytes(bins), 0), c(smmc_stats_bytes(3), 0);
  auto *ha = reinterpret_cast<smmc_stats *>(a.data());
  auto *hb = reinterpret_cast<smmc_stats *>(b.data());
  ha->n_bins = hb->n_bins = bins;
  reinterpret_cast<smmc_stats *>(c.data())->n_bins = 3;
  ha->min = INFINITY; ha->max = -INFINITY;
  hb->count = 5; hb->sum = 2.5; hb->min = -1.f; hb->max = 9.f;
  reinterpret_cast<uint64_t *>(hb + 1)[bins - 1] = 5;
  EXPECT(smmc_stats_merge(a.data(), b.data()) == SMMC_OK && ha->count == 5 && ha->min == -1.f);
  EXPECT(reinterpret_cast<uint64_t *>(ha + 1)[bins - 1] == 5);
  EXPECT(smmc_stats_merge(a.data(), c.data()) == SMMC_ERR_INVALID);
  EXPECT(smmc_stats_merge(nullptr, c.data()) == SMMC_ERR_INVALID);
  EXPECT(std::strlen(smmc_last_error()) > 0);
  smmc::Summary s;
  EXPECT(s.stddev() == 0.0);
}

static void csv_round_trip(int id) {
  const std::string path = "/tmp/smmc_san_" + std::to_string(id) + ".csv";
  {
    std::ofstream f(path);
    f << "Date,returns,other\n1928-01-01,,x\n1928-02-01,1.5,y\n1928-03-01, -2.25 ,z\n1928-04-01,nan,\n\n1928-05-01,\"3\",q\nshort\n";
  }
  std::vector<float> v = read_historical_returns(path);
  EXPECT(v.size() == 3 && v[0] == 1.5f && v[1] == -2.25f && v[2] == 3.f);
  bool threw = false;
  try { read_historical_returns("/nonexistent/smmc.csv"); } catch (const std::runtime_error &) { threw = true; }
  EXPECT(threw);
  { std::ofstream f(path); f << "a,b\n1,2\n"; }
  threw = false;
  try { read_historical_returns(path); } catch (const std::runtime_error &) { threw = true; }
  EXPECT(threw);
  std::vector<float> vals = {1000.f, 1015.f, 992.1625f};
  write_vector_file(path, vals);
  std::ifstream in(path);
  std::string text;
  std::getline(in, text);
  EXPECT(text == "1000,1015,992.162,");
  std::remove(path.c_str());
  EXPECT(smmc::bundled_synthetic_returns().size() == 1127);
}

static void sizing() {
  // the callee-sized result vector of mc_simulations_gpu: pre-faulted from two threads, then resized
  std::vector<float> big = {1.f, 2.f};
  smmc::resize_prefaulted(big, 3000001);  // 12 MB: above the threshold of the madvise path
  EXPECT(big.size() == 3000001 && big.front() == 0.f && big[1500000] == 0.f && big.back() == 0.f);
  big[7] = 5.f;
  smmc::resize_prefaulted(big, 100);  // small sizes: no madvise; always n zeros
  EXPECT(big.size() == 100 && big[7] == 0.f);
  smmc::resize_prefaulted(big, 0);
  EXPECT(big.empty());
}

static void samplers(unsigned seed) {
  std::vector<float> table = smmc::bundled_synthetic_returns();
  smmc::fix_seed(true, seed);
  std::vector<float> h = sample_returns_historical(100, table);
  EXPECT(h.size() == 100);
  std::vector<float> g = sample_returns_gaussian(1000, 0.5f, 0.8f);
  EXPECT(g.size() == 1000);
  EXPECT(sample_returns_gaussian(0, 0.f, 1.f).empty());
  smmc::fix_seed(false, 0);
  EXPECT(sample_returns_historical(3, table).size() == 3);
}

static void no_device_paths() {
  int n = -1;
  EXPECT(smmc_device_count(&n) == SMMC_OK && n >= 0);
  EXPECT(smmc_device_count(nullptr) == SMMC_ERR_INVALID);
  smmc_engine *e = reinterpret_cast<smmc_engine *>(1);
  const int rc = smmc_engine_create(0, nullptr, &e);
  if (n == 0) EXPECT(rc == SMMC_ERR_NO_DEVICE && e == nullptr);
  EXPECT(smmc_engine_create(0, nullptr, nullptr) == SMMC_ERR_INVALID);
  smmc_engine_destroy(nullptr);
  smmc_sim sim{};
  EXPECT(smmc_engine_simulate(nullptr, &sim, nullptr, nullptr, nullptr, nullptr) == SMMC_ERR_INVALID);
  EXPECT(smmc_engine_set_table(nullptr, nullptr, 0) == SMMC_ERR_INVALID);
  EXPECT(smmc_engine_set_stream(nullptr, nullptr) == SMMC_ERR_INVALID);
  EXPECT(smmc_engine_set_progress(nullptr, nullptr, nullptr) == SMMC_ERR_INVALID);
  EXPECT(smmc_engine_quartiles(nullptr, nullptr, 0, nullptr) == SMMC_ERR_INVALID);
  // the group entry: argument checks, and no device -> no group
  smmc_group *grp = reinterpret_cast<smmc_group *>(1);
  const int devs[3] = {0, 0, 1};
  EXPECT(smmc_group_create(nullptr, 1, SMMC_MERGE_HOST, &grp) == SMMC_ERR_INVALID && grp == nullptr);
  EXPECT(smmc_group_create(devs, 0, SMMC_MERGE_HOST, &grp) == SMMC_ERR_INVALID);
  EXPECT(smmc_group_create(devs, 1, 7, &grp) == SMMC_ERR_INVALID);
  EXPECT(smmc_group_create(devs, 2, SMMC_MERGE_RCCL, &grp) == SMMC_ERR_INVALID);  // device 0 twice
  EXPECT(smmc_group_create(devs, 1, SMMC_MERGE_HOST, nullptr) == SMMC_ERR_INVALID);
  if (n == 0) EXPECT(smmc_group_create(devs, 3, SMMC_MERGE_HOST, &grp) == SMMC_ERR_NO_DEVICE && grp == nullptr);
  smmc_group_destroy(nullptr);
  EXPECT(smmc_group_size(nullptr) == 0);
  EXPECT(smmc_group_simulate(nullptr, &sim, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == SMMC_ERR_INVALID);
  EXPECT(smmc_group_set_table(nullptr, nullptr, 0) == SMMC_ERR_INVALID);
  EXPECT(smmc_group_timings(nullptr, nullptr, nullptr, nullptr) == SMMC_ERR_INVALID);
  if (n != 0) return;  // the rest is the loud failure of every engine entry without a GPU
  std::vector<float> table = smmc::bundled_synthetic_returns(), out(10, 0.f), means, vars;
  std::vector<std::vector<float>> data(10);
  std::atomic<long> counter{0};
  int thrown = 0;
  try { mc_simulations(counter, 10, 5u, 1000.f, table, out); } catch (const std::runtime_error &) { ++thrown; }
  try { mc_simulations_gpu(counter, 10, 5, 1000.f, table, out, 1); } catch (const std::runtime_error &) { ++thrown; }
  try { mc_simulations_gpu(counter, 10, 5, 1000.f, table, out, 2); } catch (const std::runtime_error &) { ++thrown; }
  try { mc_simulations_gpu_reduceBlock(counter, 10, 5, 1000.f, table, means, vars, 1); } catch (const std::runtime_error &) { ++thrown; }
  try { mc_simulations_keepdata(counter, 10, 5u, 1000.f, table, data, out); } catch (const std::runtime_error &) { ++thrown; }
  try { reduce_mean_gpu(out, 10); } catch (const std::runtime_error &) { ++thrown; }
  try { smmc::mc_summary(10, 5, 1000.f, true, table, 0.5f, 0.8f, 1000.f, 4, 0.f, 1.f, 1); } catch (const std::runtime_error &) { ++thrown; }
  EXPECT(thrown == 7);
  int invalid = 0;
  try { mc_simulations_gpu_reduceBlock(counter, 10, 5, 1000.f, table, means, vars, 2); } catch (const std::invalid_argument &) { ++invalid; }
  try { mc_simulations(counter, 11, 5u, 1000.f, table, out); } catch (const std::length_error &) { ++invalid; }
  try { mc_simulations_gpu(counter, -1, 5, 1000.f, table, out, 1); } catch (const std::invalid_argument &) { ++invalid; }
  try { mc_simulations_gpu(counter, 10, 5, 1000.f, table, out, 0); } catch (const std::invalid_argument &) { ++invalid; }
  EXPECT(invalid == 4);
}

int main() {
  scalars_and_records();
  csv_round_trip(0);
  samplers(1);
  sizing();
  no_device_paths();
  // the same from four threads at once
  std::vector<std::thread> threads;
  for (int t = 0; t < 4; ++t)
    threads.emplace_back([t] {
      for (int rep = 0; rep < 3; ++rep) {
        scalars_and_records();
        csv_round_trip(t + 1);
        samplers(100 + t);
        sizing();
        no_device_paths();
      }
    });
  for (auto &t : threads) t.join();
  std::printf(fails ? "host_sanitize: %d FAILURES\n" : "host_sanitize: ok\n", fails.load());
  return fails != 0;
}
