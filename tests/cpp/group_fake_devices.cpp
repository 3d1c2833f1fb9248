// group_fake_devices.cpp -- smmc_group_* and the C++ drop-in's n_gpus calls over THREE distinct (fake) devices,
// built with ThreadSanitizer and with AddressSanitizer (tests/test_group_fake_devices_cpu.py; runtime:
// tests/cpp/fake_hip.cpp, kernels: tests/cpp/launch_fake.cpp).  The reference's counterpart is
// mc_simulations_multi_gpu_launcher_async (src/simulations.cu:576-655) behind mc_simulations_gpu(n_gpus).
//
// A fake launch writes fake_path_value(global id) where the kernel writes a final value, so "every id exactly
// once, in its place" is checkable for any sharding and chunking.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <thread>
#include <vector>

#include "smmc.h"
#include "stock_market_monte_carlo/simulations.h"

extern "C" float fake_path_value(uint64_t id, uint32_t key0, uint32_t key1, uint32_t n_periods, float capital);
extern "C" void fake_hip_fail_mallocs_on(int device);
extern "C" void fake_launch_fail_finalize(int n);
extern "C" size_t fake_hip_live_allocations(void);
extern "C" size_t fake_hip_registered_ranges(void);

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static smmc_sim make_sim(int mode, uint64_t seed, uint64_t first, uint64_t n, uint32_t periods, uint32_t bins) {
  smmc_sim s;
  std::memset(&s, 0, sizeof s);
  s.struct_size = sizeof s;
  s.mode = mode;
  s.seed = seed;
  s.first_path = first;
  s.n_paths = n;
  s.n_periods = periods;
  s.initial_capital = 1000.f;
  s.gauss_mean = 0.5f;
  s.gauss_std = 0.83333f;
  s.n_bins = bins;
  s.hist_lo = 0.f;
  s.hist_hi = 2500.f;
  s.below_threshold = 1000.f;
  return s;
}

struct Seen {
  std::atomic<int64_t> last{-1};
  std::atomic<int> calls{0};
  std::atomic<bool> monotone{true};
  static void on(void *user, int64_t finished) {
    Seen *s = static_cast<Seen *>(user);
    if (finished < s->last.load()) s->monotone = false;
    s->last = finished;
    ++s->calls;
  }
};

// the record of ids [first, first + n) computed straight from the definition
static void expected_record(const smmc_sim &s, smmc_stats *st, std::vector<uint64_t> &hist) {
  std::memset(st, 0, sizeof *st);
  st->min = INFINITY;
  st->max = -INFINITY;
  hist.assign(s.n_bins, 0);
  const double inv = s.n_bins ? double(s.n_bins) / (double(s.hist_hi) - double(s.hist_lo)) : 0.0;
  for (uint64_t i = 0; i < s.n_paths; ++i) {
    const float v = fake_path_value(s.first_path + i, uint32_t(s.seed), uint32_t(s.seed >> 32), s.n_periods, s.initial_capital);
    st->count += 1;
    st->below += v < s.below_threshold;
    st->min = std::fmin(st->min, v);
    st->max = std::fmax(st->max, v);
    if (!s.n_bins) continue;  // no histogram asked for: no bucket, no underflow / overflow count (as the kernel)
    if (v < s.hist_lo) st->underflow += 1;
    else if (v < s.hist_hi) {
      int b = int((double(v) - s.hist_lo) * inv);
      hist[b < int(s.n_bins) - 1 ? b : int(s.n_bins) - 1] += 1;
    } else st->overflow += 1;
  }
}

static bool values_in_place(const std::vector<float> &out, const smmc_sim &s) {
  for (uint64_t i = 0; i < s.n_paths; ++i)
    if (out[i] != fake_path_value(s.first_path + i, uint32_t(s.seed), uint32_t(s.seed >> 32), s.n_periods, s.initial_capital)) {
      std::printf("value %llu is not its id's\n", (unsigned long long)i);
      return false;
    }
  return true;
}

int main() {
  int n_dev = 0;  // FAKE_HIP_DEVICES: 3 (the default) or 8 (the reference's and BASELINE configs[3] / [4]'s node)
  EXPECT(smmc_device_count(&n_dev) == SMMC_OK && n_dev >= 3 && n_dev <= 8);
  const int G = n_dev;
  const int devices[8] = {0, 1, 2, 3, 4, 5, 6, 7};
  smmc_group *g = nullptr;
  EXPECT(smmc_group_create(devices, G, SMMC_MERGE_HOST, &g) == SMMC_OK && smmc_group_size(g) == G);
  if (!g) return 1;
  std::vector<float> table(1127);
  for (size_t i = 0; i < table.size(); ++i) table[i] = 0.01f * float(i) - 5.f;
  EXPECT(smmc_group_set_table(g, table.data(), uint32_t(table.size())) == SMMC_OK);
  EXPECT(smmc_group_set_table(g, table.data(), uint32_t(table.size())) == SMMC_OK);  // identical: not uploaded again

  // 1. 36 MB of final values (above the pinning threshold: the group registers the whole buffer once, its shards
  //    must not register anything), three shards with a remainder, several chunks per shard, statistics, a
  //    progress callback AND a polled counter, ids beyond 2^32
  {
    const uint64_t n = 9000001;
    smmc_sim s = make_sim(SMMC_MODE_TABLE, 0x1234567ull << 20, (1ull << 33) + 5, n, 360, 64);
    std::vector<float> out(n + 2, -7.f);  // out + 1: not page aligned
    smmc_stats st;
    std::vector<uint64_t> hist(64), want_hist;
    Seen seen;
    volatile int64_t counter = -1;
    EXPECT(smmc_group_set_progress(g, &Seen::on, &seen) == SMMC_OK);
    std::atomic<bool> stop{false};
    std::atomic<int> poll_steps{0};
    std::thread poller([&] {  // a GUI thread polling the counter while the engines run (visualize_returns_cpu_v2.cpp:360-376)
      int64_t prev = -1;
      while (!stop) {
        const int64_t c = __atomic_load_n(const_cast<int64_t *>(&counter), __ATOMIC_ACQUIRE);
        if (c != prev) ++poll_steps;
        prev = c;
        std::this_thread::yield();
      }
    });
    const int rc = smmc_group_simulate(g, &s, out.data() + 1, nullptr, nullptr, &counter, &st, hist.data());
    stop = true;
    poller.join();
    EXPECT(rc == SMMC_OK);
    if (rc != SMMC_OK) std::printf("%s\n", smmc_last_error());
    EXPECT(out[0] == -7.f && out[n + 1] == -7.f);
    std::vector<float> body(out.begin() + 1, out.begin() + 1 + n);
    EXPECT(values_in_place(body, s));
    smmc_stats want;
    expected_record(s, &want, want_hist);
    EXPECT(st.count == n && st.below == want.below && st.underflow == want.underflow && st.overflow == want.overflow);
    EXPECT(st.min == want.min && st.max == want.max && hist == want_hist && st.n_bins == 64);
    EXPECT(seen.monotone && seen.last == int64_t(n) && seen.calls >= 6 && counter == int64_t(n) && poll_steps >= 2);
    EXPECT(fake_hip_registered_ranges() == 0);  // the group's registration is released again
    (void)smmc_group_set_progress(g, nullptr, nullptr);
    uint64_t first, count, total = 0;
    for (int i = 0; i < G; ++i) {
      EXPECT(smmc_group_shard(g, n, i, &first, &count) == SMMC_OK && first == total);
      total += count;
    }
    EXPECT(total == n);
  }

  // 2. chunk means / variances need shards that start on a multiple of 256 paths; Gaussian mode needs no table
  {
    const uint64_t n = uint64_t(G) * 256 * 700;
    smmc_sim s = make_sim(SMMC_MODE_GAUSSIAN, 99, 0, n, 36, 0);
    std::vector<float> out(n), cm(n / 256), cv(n / 256);
    EXPECT(smmc_group_simulate(g, &s, out.data(), cm.data(), cv.data(), nullptr, nullptr, nullptr) == SMMC_OK);
    EXPECT(values_in_place(out, s));
    bool ok = true;
    for (uint64_t c = 0; c < n / 256; ++c) {
      double s1 = 0, s2 = 0;
      for (int j = 0; j < 256; ++j) { const double v = out[c * 256 + j]; s1 += v; s2 += v * v; }
      const double mean = s1 / 256, var = s2 / 256 - mean * mean;
      ok = ok && cm[c] == float(mean) && cv[c] == float(var > 0 ? var : 0);
    }
    EXPECT(ok);
    smmc_sim odd = make_sim(SMMC_MODE_GAUSSIAN, 99, 0, n + 1, 36, 0);  // shard 1 would start at 179201 (G = 3)
    std::vector<float> out2(n + 1);
    EXPECT(smmc_group_simulate(g, &odd, out2.data(), cm.data(), nullptr, nullptr, nullptr, nullptr) == SMMC_ERR_INVALID);
  }

  // 3. two callers at once on two groups over the same devices (the reference's GUI runs two engines side by
  //    side, examples/visualize_returns_cpu_v2.cpp:185-202)
  {
    smmc_group *g2 = nullptr;
    EXPECT(smmc_group_create(devices, G, SMMC_MERGE_HOST, &g2) == SMMC_OK);
    EXPECT(smmc_group_set_table(g2, table.data(), uint32_t(table.size())) == SMMC_OK);
    const uint64_t n = 1500007;
    smmc_sim sa = make_sim(SMMC_MODE_TABLE, 7, 0, n, 360, 16), sb = make_sim(SMMC_MODE_TABLE, 8, 1000, n, 36, 16);
    std::vector<float> oa(n), ob(n);
    smmc_stats sta, stb;
    std::vector<uint64_t> ha(16), hb(16);
    int rca = -1, rcb = -1;
    std::thread ta([&] { rca = smmc_group_simulate(g, &sa, oa.data(), nullptr, nullptr, nullptr, &sta, ha.data()); });
    std::thread tb([&] { rcb = smmc_group_simulate(g2, &sb, ob.data(), nullptr, nullptr, nullptr, &stb, hb.data()); });
    ta.join();
    tb.join();
    EXPECT(rca == SMMC_OK && rcb == SMMC_OK && values_in_place(oa, sa) && values_in_place(ob, sb));
    EXPECT(sta.count == n && stb.count == n);
    smmc_group_destroy(g2);
  }

  // 4. a set_table that fails on the second device: the devices hold different tables, table mode is refused
  //    until a set_table has succeeded everywhere; Gaussian mode is not affected
  {
    std::vector<float> other(table);
    other[0] += 1.f;
    fake_hip_fail_mallocs_on(1);
    std::vector<float> longer(2000, 1.f);  // another length: the engine has to allocate
    EXPECT(smmc_group_set_table(g, longer.data(), uint32_t(longer.size())) != SMMC_OK);
    EXPECT(std::strstr(smmc_last_error(), "different tables") != nullptr);
    fake_hip_fail_mallocs_on(-1);
    smmc_sim s = make_sim(SMMC_MODE_TABLE, 7, 0, 3000, 36, 0);
    std::vector<float> out(3000);
    EXPECT(smmc_group_simulate(g, &s, out.data(), nullptr, nullptr, nullptr, nullptr, nullptr) == SMMC_ERR_INVALID);
    smmc_sim sg = make_sim(SMMC_MODE_GAUSSIAN, 7, 0, 3000, 36, 0);
    EXPECT(smmc_group_simulate(g, &sg, out.data(), nullptr, nullptr, nullptr, nullptr, nullptr) == SMMC_OK);
    EXPECT(smmc_group_set_table(g, other.data(), uint32_t(other.size())) == SMMC_OK);
    EXPECT(smmc_group_simulate(g, &s, out.data(), nullptr, nullptr, nullptr, nullptr, nullptr) == SMMC_OK && values_in_place(out, s));
  }
  smmc_group_destroy(g);

  // 5. the C++ drop-in's n_gpus calls (mc_simulations_gpu :661-680 of the reference): three devices, the callee
  //    sizes the result, the remainder is kept, the counter ends at N
  {
    smmc::fix_seed(true, 4242);
    std::atomic<long> counter{0};
    std::vector<float> totals;
    const long n = 9000003;
    mc_simulations_gpu(counter, n, 360, 1000.f, table, totals, G);
    EXPECT(long(totals.size()) == n && counter == n);
    smmc_sim s = make_sim(SMMC_MODE_TABLE, 4242, 0, uint64_t(n), 360, 0);
    EXPECT(values_in_place(totals, s));
    smmc::Summary sum = smmc::mc_summary(n, 360, 1000.f, false, table, 0.f, 0.f, 1000.f, 64, 0.f, 2500.f, G);
    smmc_sim sh = make_sim(SMMC_MODE_TABLE, 4242, 0, uint64_t(n), 360, 64);
    smmc_stats want;
    std::vector<uint64_t> want_hist;
    expected_record(sh, &want, want_hist);
    EXPECT(sum.count == uint64_t(n) && sum.below == want.below && sum.hist == want_hist && sum.min == want.min && sum.max == want.max);
    bool threw = false;
    try { mc_simulations_gpu(counter, 10, 36, 1000.f, table, totals, G + 1); } catch (const std::invalid_argument &) { threw = true; }
    EXPECT(threw);  // one shard more than there are devices
  }
  // 5b. the bucket accumulator of an engine is zero between launches because finalize_kernel leaves it so (no memset per
  //     call); an enqueue that fails between a kernel and its fold leaves counts behind, and the NEXT call must clear
  //     them first: the record after the failure is the record of that call alone, not of one and a half
  {
    smmc_engine *e = nullptr;
    EXPECT(smmc_engine_create(0, nullptr, &e) == SMMC_OK);
    if (e) {
      EXPECT(smmc_engine_set_table(e, table.data(), uint32_t(table.size())) == SMMC_OK);
      const uint64_t n = 700001;
      smmc_sim s = make_sim(SMMC_MODE_TABLE, 77, 5, n, 360, 64);
      std::vector<float> out(n);
      std::vector<uint64_t> hist(64), want_hist;
      smmc_stats st, want;
      expected_record(s, &want, want_hist);
      EXPECT(smmc_engine_simulate_to_host(e, &s, out.data(), nullptr, nullptr, nullptr, &st, hist.data()) == SMMC_OK && hist == want_hist);
      fake_launch_fail_finalize(1);
      EXPECT(smmc_engine_simulate_to_host(e, &s, out.data(), nullptr, nullptr, nullptr, &st, hist.data()) != SMMC_OK);
      fake_launch_fail_finalize(0);
      std::fill(hist.begin(), hist.end(), 5);
      EXPECT(smmc_engine_simulate_to_host(e, &s, out.data(), nullptr, nullptr, nullptr, &st, hist.data()) == SMMC_OK);
      EXPECT(hist == want_hist && st.count == n && st.below == want.below);
      EXPECT(smmc_engine_simulate_to_host(e, &s, out.data(), nullptr, nullptr, nullptr, &st, hist.data()) == SMMC_OK && hist == want_hist);
      smmc_engine_destroy(e);
    }
  }

  // 6. the RCCL merge with G distinct devices (FAKE_RCCL=1: tests/cpp/fake_rccl.cpp is the librccl.so.1 on the library
  //    path -- host memory, completes inside ncclGroupEnd, refuses ranks that disagree).  What real RCCL would show as a
  //    hang: a rank missing from the bracket, unequal counts; what it would show as garbage: a wrong offset or type.
  if (std::getenv("FAKE_RCCL")) {
    void *lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    typedef long (*counter_fn)(void);
    typedef void (*fail_fn)(long);
    counter_fn collectives = lib ? reinterpret_cast<counter_fn>(dlsym(lib, "fake_rccl_collectives")) : nullptr;
    counter_fn brackets = lib ? reinterpret_cast<counter_fn>(dlsym(lib, "fake_rccl_brackets")) : nullptr;
    counter_fn live = lib ? reinterpret_cast<counter_fn>(dlsym(lib, "fake_rccl_live_comms")) : nullptr;
    fail_fn fail_after = lib ? reinterpret_cast<fail_fn>(dlsym(lib, "fake_rccl_fail_after")) : nullptr;
    EXPECT(collectives && brackets && live && fail_after);  // not the fake: the real librccl must never meet the fake runtime
    if (collectives && brackets && live && fail_after) {
      smmc_group *gr = nullptr, *gh = nullptr;
      EXPECT(smmc_group_create(devices, G, SMMC_MERGE_RCCL, &gr) == SMMC_OK && live() == G);
      EXPECT(smmc_group_create(devices, G, SMMC_MERGE_HOST, &gh) == SMMC_OK);
      if (gr && gh) {
        EXPECT(smmc_group_set_table(gr, table.data(), uint32_t(table.size())) == SMMC_OK);
        EXPECT(smmc_group_set_table(gh, table.data(), uint32_t(table.size())) == SMMC_OK);
        double engines_ms = -1, comm_ms = -1, merge_ms = -1;
        EXPECT(smmc_group_timings(gr, &engines_ms, &comm_ms, &merge_ms) == SMMC_OK && comm_ms >= 0.0);
        for (uint32_t bins : {64u, 0u, 1000u}) {
          const uint64_t n = 2000003;
          smmc_sim s = make_sim(SMMC_MODE_TABLE, 31 + bins, (1ull << 40) + 7, n, 360, bins);
          smmc_stats sr, sh;
          std::vector<uint64_t> hr(bins + 1, 77), hh(bins + 1, 77), want_hist;
          const long c0 = collectives(), b0 = brackets();
          const int rc = smmc_group_simulate(gr, &s, nullptr, nullptr, nullptr, nullptr, &sr, hr.data());
          if (rc != SMMC_OK) std::printf("%s\n", smmc_last_error());
          EXPECT(rc == SMMC_OK);
          EXPECT(collectives() - c0 == (bins ? 2 : 1) && brackets() - b0 == 1);  // ONE bracket: the counters, the buckets
          EXPECT(smmc_group_simulate(gh, &s, nullptr, nullptr, nullptr, nullptr, &sh, hh.data()) == SMMC_OK);
          EXPECT(std::memcmp(&sr, &sh, sizeof sr) == 0 && hr == hh && hr[bins] == 77);  // the same bits as the host merge
          smmc_stats want;
          expected_record(s, &want, want_hist);
          hr.resize(bins);
          EXPECT(sr.count == n && sr.below == want.below && sr.underflow == want.underflow && sr.overflow == want.overflow);
          EXPECT(sr.min == want.min && sr.max == want.max && hr == want_hist && sr.n_bins == bins);
        }
        // a collective that fails on the second device: an error with RCCL's words, nothing left queued, and the next call works
        smmc_sim s = make_sim(SMMC_MODE_GAUSSIAN, 5, 0, 300001, 36, 16);
        smmc_stats sr, sh;
        std::vector<uint64_t> hr(16), hh(16);
        fail_after(2);  // device 0's two all-reduces are posted, device 1's first one fails
        EXPECT(smmc_group_simulate(gr, &s, nullptr, nullptr, nullptr, nullptr, &sr, hr.data()) == SMMC_ERR_HIP);
        EXPECT(std::strstr(smmc_last_error(), "RCCL all-reduce") != nullptr && std::strstr(smmc_last_error(), "injected") != nullptr);
        EXPECT(smmc_group_simulate(gr, &s, nullptr, nullptr, nullptr, nullptr, &sr, hr.data()) == SMMC_OK);
        EXPECT(smmc_group_simulate(gh, &s, nullptr, nullptr, nullptr, nullptr, &sh, hh.data()) == SMMC_OK);
        EXPECT(std::memcmp(&sr, &sh, sizeof sr) == 0 && hr == hh);
      }
      smmc_group_destroy(gr);
      smmc_group_destroy(gh);
      EXPECT(live() == 0);
      // the same device twice is not a clique RCCL accepts: refused by smmc_group_create, with a reason
      const int twice[2] = {0, 0};
      smmc_group *bad = nullptr;
      EXPECT(smmc_group_create(twice, 2, SMMC_MERGE_RCCL, &bad) != SMMC_OK && bad == nullptr);
      // the C++ drop-in with SMMC_GROUP_MERGE=rccl: the summary of the n_gpus call, merged by the all-reduce
      smmc::fix_seed(true, 4242);
      const long n = 3000001;
      smmc::Summary by_host = smmc::mc_summary(n, 360, 1000.f, false, table, 0.f, 0.f, 1000.f, 64, 0.f, 2500.f, G);
      setenv("SMMC_GROUP_MERGE", "rccl", 1);
      const long c0 = collectives();
      smmc::Summary by_rccl = smmc::mc_summary(n, 360, 1000.f, false, table, 0.f, 0.f, 1000.f, 64, 0.f, 2500.f, G);
      unsetenv("SMMC_GROUP_MERGE");
      EXPECT(collectives() - c0 >= 2);
      EXPECT(by_rccl.count == by_host.count && by_rccl.below == by_host.below && by_rccl.hist == by_host.hist);
      EXPECT(by_rccl.min == by_host.min && by_rccl.max == by_host.max && by_rccl.sum == by_host.sum && by_rccl.sumsq == by_host.sumsq);
      std::printf("rccl merge over %d fake devices: ok so far (%ld collectives)\n", G, collectives());
    }
  }
  if (fake_hip_registered_ranges() != 0) { std::printf("FAIL: %zu host ranges still registered\n", fake_hip_registered_ranges()); ++fails; }
  std::printf(fails ? "group_fake_devices: %d failure(s)\n" : "group_fake_devices: ok\n", fails);
  return fails ? 1 : 0;
}
