// host_check.cpp -- the reference-named functions that need no GPU (src/simulations.cpp:14-55,
// 83-112), called through the drop-in library; prints one JSON object for tests/test_samplers_cpu.py.
#include <cstdio>
#include <cstdlib>

#include "stock_market_monte_carlo/simulations.h"

static void print_vec(const char *name, const std::vector<float> &v, const char *tail) {
  std::printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); ++i) std::printf("%s%.9g", i ? ", " : "", v[i]);
  std::printf("]%s", tail);
}

int main(int argc, char **argv) {
  const char *csv = argc > 1 ? argv[1] : "data/SP500_monthly_returns.csv";
  std::vector<float> table = read_historical_returns(csv);
  std::printf("{\"table_len\": %zu, ", table.size());
  // sample_returns_historical (src/simulations.cpp:95-112): mt19937(seed) + uniform_int_distribution
  for (unsigned seed : {0u, 1000u, 1001u, 4294967295u}) {
    smmc::fix_seed(true, seed);
    char name[32];
    std::snprintf(name, sizeof name, "hist_%u", seed);
    print_vec(name, sample_returns_historical(48, table), ", ");
  }
  // sample_returns_gaussian (src/simulations.cpp:41-55): N(mean, std)
  smmc::fix_seed(true, 2024);
  const unsigned n = 400000;
  std::vector<float> g = sample_returns_gaussian(n, 0.5f, 0.83333f);
  double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  for (float v : g) s1 += v;
  const double mean = s1 / n;
  for (float v : g) {
    const double d = v - mean;
    s2 += d * d;
    s3 += d * d * d;
    s4 += d * d * d * d;
  }
  std::vector<float> g2 = sample_returns_gaussian(n, 0.5f, 0.83333f);  // fixed seed: the same draws again
  smmc::fix_seed(false, 0);
  std::vector<float> g3 = sample_returns_gaussian(8, 0.5f, 0.83333f);  // unseeded: the reference's behaviour
  bool threw = false;
  try {
    std::vector<float> empty;
    sample_returns_historical(3, empty);
  } catch (const std::out_of_range &) {  // .at() in the reference, src/simulations.cpp:108
    threw = true;
  }
  std::printf("\"gauss_n\": %u, \"gauss_size\": %zu, \"gauss_mean\": %.17g, \"gauss_var\": %.17g, \"gauss_m3\": %.17g, "
              "\"gauss_m4\": %.17g, \"gauss_repeat\": %s, \"unseeded_size\": %zu, \"empty_threw\": %s, ",
              n, g.size(), mean, s2 / n, s3 / n, s4 / n, g == g2 ? "true" : "false", g3.size(), threw ? "true" : "false");
  std::vector<float> rets = {1.f, -2.f, 3.5f};
  print_vec("mu", many_updates(1000.f, rets, 3u), ", ");
  std::printf("\"update_fund\": %.9g}\n", update_fund(1000.f, 0.5f));
  return 0;
}
