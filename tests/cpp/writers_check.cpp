// writers_check.cpp -- calls the drop-in's CSV writers (host-only) for tests/test_ref_callers_cpu.py
#include "stock_market_monte_carlo/simulations.h"
int main() {
  std::vector<float> r = {1.5f, -2.25f, 0.1f, 1e-7f, 12345.678f, -0.0f};
  std::vector<float> v = {1000.f, 1015.f, 992.1625f, 993.154663f, 1.0e9f, 3.4e38f, 1.17549435e-38f};
  write_data_file("ours.csv", r, v);
  write_vector_file("outputs/ours_vec.csv", v);
  return 0;
}
