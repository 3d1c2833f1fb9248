// asref_driver.cpp -- runs oracle/asref_cpu.cpp (the reference's per-path std::random_device loop)
// under ASan/UBSan: both seeding modes, ragged block counts, one-entry table, zero paths.
#include <cstdint>
#include <cstdio>
#include <vector>

extern "C" int orc_asref_mc_simulations(int64_t, uint32_t, float, const float *, uint32_t, float *, int, int, uint32_t);
extern "C" int orc_asref_gaussian_mc(int64_t, uint32_t, float, float, float, uint32_t, float *, int);

int main() {
  std::vector<float> table(1127);
  for (size_t i = 0; i < table.size(); ++i) table[i] = float(i % 23) - 9.f;
  std::vector<float> out(2501, -1.f);
  int fails = 0;
  fails += orc_asref_mc_simulations(2501, 37, 1000.f, table.data(), 1127, out.data(), 2, 0, 0) != 2;
  fails += orc_asref_mc_simulations(2501, 37, 1000.f, table.data(), 1127, out.data(), 3, 1, 0xfffffff0u) != 3;
  fails += orc_asref_mc_simulations(0, 37, 1000.f, table.data(), 1127, out.data(), 1, 0, 0) != 1;
  fails += orc_asref_mc_simulations(7, 0, 1000.f, table.data(), 1, out.data(), 1, 1, 5) != 1 || out[6] != 1000.f;
  fails += orc_asref_gaussian_mc(2501, 37, 1000.f, 0.5f, 0.83333f, 0xfffffff0u, out.data(), 3) != 3 || !(out[2500] > 0.f);
  fails += orc_asref_gaussian_mc(0, 37, 1000.f, 0.5f, 0.83333f, 1, out.data(), 1) != 1;
  fails += orc_asref_gaussian_mc(7, 0, 1000.f, 0.5f, 0.83333f, 1, out.data(), 1) != 1 || out[6] != 1000.f;
  std::printf(fails ? "asref_driver: FAILURES\n" : "asref_driver: ok\n");
  return fails != 0;
}
