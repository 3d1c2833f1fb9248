// pin_libstdcxx.cpp -- golden-vector generator (test infrastructure).
//
// The reference draws table indices with std::mt19937 +
// std::uniform_int_distribution<int> (src/simulations.cpp:245-250).  Those live
// in libstdc++, not in the reference tree.  This program runs the SYSTEM
// libstdc++ (GCC 11.4 here) on fixed seeds and prints JSON that
// tests/golden/make_golden.py stores under tests/golden/; the C oracle's
// hand-written mt19937 + Lemire map must reproduce it exactly.
//
// It is NOT the reference: the per-path loop below is a second, independent
// restatement (with the real library distributions) of src/simulations.cpp:
// 240-252 with an explicit seed where the reference uses std::random_device.
//
// usage: pin_libstdcxx <table.txt>   (one float per line, percent units)
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>
#include <vector>

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

int main(int argc, char **argv) {
  if (argc != 2) { std::fprintf(stderr, "usage: %s table.txt\n", argv[0]); return 2; }
  std::vector<float> table;
  { std::ifstream in(argv[1]); float v; while (in >> v) table.push_back(v); }
  const uint32_t seeds[] = {0u, 1u, 1000u, 1001u, 5489u, 123456789u, 0x7fffffffu, 0xffffffffu};
  const int ranges[] = {1127, 1, 2, 3, 1000, 65536, 1000003, 2147483647};

  std::printf("{\n \"libstdcxx\": \"%d\",\n \"table_len\": %zu,\n", __GLIBCXX__, table.size());
  // raw engine outputs
  std::printf(" \"mt19937_raw\": [\n");
  for (size_t s = 0; s < 8; s++) {
    std::mt19937 rng(seeds[s]);
    std::printf("  {\"seed\": %u, \"out\": [", seeds[s]);
    for (int i = 0; i < 8; i++) std::printf("%s%lu", i ? "," : "", (unsigned long)rng());
    std::printf("]}%s\n", s + 1 < 8 ? "," : "");
  }
  std::printf(" ],\n \"mt19937_default_10000th\": ");
  { std::mt19937 rng; unsigned long v = 0; for (int i = 0; i < 10000; i++) v = rng(); std::printf("%lu,\n", v); }
  // distribution outputs
  std::printf(" \"uniform_int\": [\n");
  bool first = true;
  for (size_t s = 0; s < 8; s++)
    for (size_t r = 0; r < 8; r++) {
      std::mt19937 rng(seeds[s]);
      std::uniform_int_distribution<int> uni(0, ranges[r] - 1);
      std::printf("%s  {\"seed\": %u, \"range\": %d, \"out\": [", first ? "" : ",\n", seeds[s], ranges[r]);
      first = false;
      for (int i = 0; i < 48; i++) std::printf("%s%d", i ? "," : "", uni(rng));
      std::printf("]}");
    }
  std::printf("\n ],\n");
  // whole paths: explicit seed, table draw, fund * (100 + r) / 100 in binary32
  std::printf(" \"paths\": [\n");
  // 226 / 227 / 228, 454 / 455, 623 / 624 / 625, 681, 850, 908, 1077 / 1078, 1135, 1304, 1531, 1816 / 1817: lengths at which
  // the device kernels change how they obtain the generator's state words -- the stretches of ref_tree_kernel, the
  // hand-over from its short instantiation to its long one and from that to ref_generic_kernel
  // (tests/test_ref_stream_gpu.py compares the HIP result with these, not with the oracle)
  const unsigned periods[] = {1, 4, 226, 227, 228, 360, 454, 455, 623, 624, 625, 681, 850, 908, 1000, 1077, 1078, 1135, 1304, 1531,
                              1816, 1817, 2000};
  first = true;
  for (unsigned P : periods)
    for (uint32_t seed0 : {1000u, 4000000000u}) {
      std::printf("%s  {\"n_periods\": %u, \"seed0\": %u, \"initial_capital\": 1000.0, \"final_bits\": [",
                  first ? "" : ",\n", P, seed0);
      first = false;
      for (uint32_t id = 0; id < 32; id++) {
        std::mt19937 rng(uint32_t(seed0 + id));
        std::uniform_int_distribution<int> uni(0, int(table.size()) - 1);
        float total = 1000.0f;
        for (unsigned i = 0; i < P; i++) {
          float r = table[uni(rng)];
          float a = 100.0f + r;
          float m = total * a;
          total = m / 100.0f;
        }
        std::printf("%s%u", id ? "," : "", bits(total));
      }
      std::printf("]}");
    }
  std::printf("\n ],\n");
  // paths in which the real uniform_int_distribution REJECTS a generator output (Lemire's method: 2.6e-7 per draw
  // for 1127 entries): found by counting the engine calls a path makes.  They pin what happens after a rejection
  // -- every later draw comes from the next output -- to the library itself.
  struct Counting {
    std::mt19937 g;
    unsigned long calls = 0;
    using result_type = std::mt19937::result_type;
    static constexpr result_type min() { return std::mt19937::min(); }
    static constexpr result_type max() { return std::mt19937::max(); }
    explicit Counting(uint32_t seed) : g(seed) {}
    result_type operator()() { ++calls; return g(); }
  };
  std::printf(" \"rejecting_paths\": [\n");
  first = true;
  for (unsigned P : {360u, 1000u}) {
    int found = 0;
    for (uint32_t seed = 0; seed < 4000000u && found < 6; seed++) {
      Counting rng(seed);
      std::uniform_int_distribution<int> uni(0, int(table.size()) - 1);
      float total = 1000.0f;
      for (unsigned i = 0; i < P; i++) {
        float a = 100.0f + table[uni(rng)];
        float m = total * a;
        total = m / 100.0f;
      }
      if (rng.calls == P) continue;
      std::printf("%s  {\"n_periods\": %u, \"seed\": %u, \"engine_calls\": %lu, \"initial_capital\": 1000.0, \"final_bits\": %u}",
                  first ? "" : ",\n", P, seed, rng.calls, bits(total));
      first = false;
      found++;
    }
  }
  std::printf("\n ],\n");
  // whole trajectories as mc_simulations_keepdata forms them (src/simulations.cpp:175-186): the returns of
  // sample_returns_historical (:95-112: mt19937 + uniform_int_distribution, .at(idx)), then many_updates
  std::printf(" \"trajectories\": [\n");
  first = true;
  for (unsigned P : {40u, 360u, 1000u})  // 1000: ref_tree_kernel's trajectories
    for (uint32_t seed : {1000u, 4294967295u, 32569u}) {  // 32569: a path that rejects an output within 360 draws
      std::mt19937 rng(seed);
      std::uniform_int_distribution<int> uni(0, int(table.size()) - 1);
      std::vector<float> returns;
      for (unsigned i = 0; i < P; i++) returns.push_back(table.at(uni(rng)));
      std::printf("%s  {\"n_periods\": %u, \"seed\": %u, \"initial_capital\": 1000.0, \"value_bits\": [%u", first ? "" : ",\n", P, seed,
                  bits(1000.0f));
      first = false;
      float total = 1000.0f;
      for (unsigned i = 0; i < P; i++) {
        float a = 100.0f + returns[i];
        float m = total * a;
        total = m / 100.0f;
        std::printf(",%u", bits(total));
      }
      std::printf("]}");
    }
  std::printf("\n ]\n}\n");
  return 0;
}
