// smmc_dropin.cpp -- the reference's C++ free functions (declared in
// include/stock_market_monte_carlo/simulations.h) implemented over the C ABI.
//
// Host-side counterpart of the parts of src/simulations.cpp and the wrappers at
// src/simulations.cu:661-697 that callers link against.  No HIP types appear here:
// everything device-side goes through smmc.h.
#include "stock_market_monte_carlo/simulations.h"

#include <chrono>
#include <cmath>
#include <ctime>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <filesystem>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <sstream>
#include <stdexcept>
#include <thread>
#include <utility>

#include <sys/mman.h>
#include <unistd.h>

#include "smmc.h"
#include "smmc_host.h"
#include "stock_market_monte_carlo/gpu.h"

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23  // Linux 5.14
#endif

namespace {

[[noreturn]] void raise(int rc) {
  std::string msg = std::string("smmc: ") + smmc_last_error();
  if (rc == SMMC_ERR_INVALID) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}
void check(int rc) {
  if (rc != SMMC_OK) raise(rc);
}

// One engine per device for the process (single-device entry points); an engine is not re-entrant, so
// calls on the same device serialise on its mutex (calls on different devices run concurrently).
struct Slot {
  std::mutex busy;
  smmc_engine *engine = nullptr;
  ~Slot() {
    if (engine) smmc_engine_destroy(engine);
  }
};
std::mutex g_slots_mutex;
std::map<int, std::unique_ptr<Slot>> g_slots;

Slot &slot_for(int device) {
  std::lock_guard<std::mutex> lock(g_slots_mutex);
  auto &s = g_slots[device];
  if (!s) s.reset(new Slot());
  return *s;
}

// One group (smmc_group_*, include/smmc.h) per device list and merge back-end for the process: the
// n_gpus-way entry points.  Its engines -- and, with SMMC_GROUP_MERGE=rccl, its RCCL communicator --
// are created once and kept.
struct GroupSlot {
  std::mutex busy;
  smmc_group *group = nullptr;
  ~GroupSlot() {
    if (group) smmc_group_destroy(group);
  }
};
std::mutex g_groups_mutex;
std::map<std::pair<std::vector<int>, int>, std::unique_ptr<GroupSlot>> g_groups;

GroupSlot &group_slot_for(const std::vector<int> &devices, int merge) {
  std::lock_guard<std::mutex> lock(g_groups_mutex);
  auto &s = g_groups[{devices, merge}];
  if (!s) s.reset(new GroupSlot());
  return *s;
}

std::mutex g_seed_mutex;
bool g_seed_fixed = false;
std::uint64_t g_seed = 0;

std::uint64_t next_seed() {
  {
    std::lock_guard<std::mutex> lock(g_seed_mutex);
    if (g_seed_fixed) return g_seed;
  }
  if (const char *env = std::getenv("SMMC_SEED")) return std::strtoull(env, nullptr, 0);
  std::random_device rd;  // the reference's entropy source, src/simulations.cpp:245
  return (static_cast<std::uint64_t>(rd()) << 32) ^ rd();
}

smmc_sim make_sim(int mode, std::uint64_t seed, std::uint64_t first, std::uint64_t n, unsigned periods, float capital) {
  smmc_sim s{};
  s.struct_size = sizeof(smmc_sim);
  s.mode = mode;
  s.seed = seed;
  s.first_path = first;
  s.n_paths = n;
  s.n_periods = periods;
  s.initial_capital = capital;
  s.gauss_mean = 0.5f;
  s.gauss_std = 0.83333f;
  s.below_threshold = capital;
  // SMMC_STREAM=2: counter stream v2 (round 1) for callers that hold v2 results.  SMMC_STREAM=ref: the
  // reference CPU engine's own stream for the table-draw engines (src/simulations.cpp:240-252: path id draws
  // from std::mt19937(seed + id) through uniform_int_distribution) -- with SMMC_SEED / fix_seed the final
  // values are what the reference's mc_simulations computes when its generators are seeded that way
  if (const char *env = std::getenv("SMMC_STREAM")) {
    if (env[0] == '2' && env[1] == '\0') s.flags |= SMMC_FLAG_STREAM_V2;
    if (!std::strcmp(env, "ref") && mode == SMMC_MODE_TABLE) s.flags |= SMMC_FLAG_STREAM_REF;
  }
  return s;
}

int visible_devices() {
  int n = 0;
  check(smmc_device_count(&n));
  return n;
}

// SMMC_DEVICE_MAP="0,0,1": shard g of an n_gpus-way call runs on device map[g] (several shards may
// share a device; each then has its own engine and stream).  Without it shard g runs on device g, as
// the reference's cudaSetDevice(k) loop does (src/simulations.cu:599-626).  Lets the multi-shard code
// path -- thread fan-out, per-shard engines, host merge -- run on a box with fewer GPUs than n_gpus.
std::vector<int> device_map(int n_gpus) {
  std::vector<int> map;
  if (const char *env = std::getenv("SMMC_DEVICE_MAP")) {
    std::stringstream ss(env);
    std::string cell;
    while (std::getline(ss, cell, ',')) {
      char *end = nullptr;
      const long v = std::strtol(cell.c_str(), &end, 10);
      if (end == cell.c_str() || v < 0) throw std::invalid_argument("smmc: SMMC_DEVICE_MAP must be a comma-separated list of device ids");
      map.push_back(static_cast<int>(v));
    }
    if (static_cast<int>(map.size()) < n_gpus)
      throw std::invalid_argument("smmc: SMMC_DEVICE_MAP names fewer devices than n_gpus");
    map.resize(n_gpus);
    return map;
  }
  for (int g = 0; g < n_gpus; ++g) map.push_back(g);
  return map;
}

// The group an n_gpus-way call runs on, locked for the call, with `table` loaded (table may be null for
// Gaussian runs).  Shard g covers floor(N/G) paths plus one of the N mod G leftovers (the reference
// drops the remainder, src/simulations.cu:602-603); it runs on device g, or device map[g] under
// SMMC_DEVICE_MAP.  The per-device statistics records are merged on the host in shard order, or --
// SMMC_GROUP_MERGE=rccl, distinct devices -- by one RCCL all-reduce (both give the same bits).
struct GroupSession {
  std::unique_lock<std::mutex> lock;
  smmc_group *group = nullptr;
  GroupSession(int n_gpus, const std::vector<float> *table) {
    if (n_gpus < 1) throw std::invalid_argument("smmc: n_gpus must be >= 1");
    const int have = visible_devices();
    if (have == 0) throw std::runtime_error("smmc: no MI355X visible; this library has no CPU fallback");
    const std::vector<int> map = device_map(n_gpus);
    bool distinct = true;
    for (size_t a = 0; a < map.size(); ++a) {
      if (map[a] >= have) throw std::invalid_argument("smmc: n_gpus exceeds the visible devices");
      for (size_t b = a + 1; b < map.size(); ++b) distinct = distinct && map[a] != map[b];
    }
    int merge = SMMC_MERGE_HOST;
    if (const char *env = std::getenv("SMMC_GROUP_MERGE"))
      if (!std::strcmp(env, "rccl") && distinct) merge = SMMC_MERGE_RCCL;
    GroupSlot &slot = group_slot_for(map, merge);
    lock = std::unique_lock<std::mutex>(slot.busy);
    if (!slot.group) check(smmc_group_create(map.data(), n_gpus, merge, &slot.group));
    group = slot.group;
    if (table) {
      if (table->empty()) throw std::invalid_argument("smmc: empty returns table");
      check(smmc_group_set_table(group, table->data(), static_cast<uint32_t>(table->size())));
    }
  }
};

// Engine of `device` with `table` loaded (table may be null): the single-device entry points.
struct Session {
  std::unique_lock<std::mutex> lock;
  smmc_engine *engine;
  Session(int device, const std::vector<float> *table) : lock(slot_for(device).busy) {
    Slot &s = slot_for(device);
    if (!s.engine) check(smmc_engine_create(device, SMMC_STREAM_NEW, &s.engine));
    engine = s.engine;
    if (table) {
      if (table->empty()) throw std::invalid_argument("smmc: empty returns table");
      check(smmc_engine_set_table(engine, table->data(), static_cast<uint32_t>(table->size())));
    }
  }
};

// Progress of a run: the group reports the number of finished paths over all its devices
// (smmc_group_set_progress); it goes to the caller's atomic counter (src/simulations.cpp:254).
struct Progress {
  std::atomic<long> *counter;
  static void on_progress(void *user, int64_t finished) {
    Progress *p = static_cast<Progress *>(user);
    long cur = p->counter->load(std::memory_order_relaxed);
    const long now = static_cast<long>(finished);
    while (now > cur && !p->counter->compare_exchange_weak(cur, now, std::memory_order_release)) {
    }
  }
};

double seconds_since(const std::chrono::steady_clock::time_point &t0) {
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
bool verbose() {
  const char *env = std::getenv("SMMC_VERBOSE");
  return env && *env && *env != '0';
}

void run_final_values(std::atomic<long> &n_simulations, long n_total, unsigned periods, float capital, int mode,
                      const std::vector<float> *table, float mean, float stddev, float *out, int n_gpus) {
  const std::uint64_t seed = next_seed();
  n_simulations = 0;
  GroupSession gs(n_gpus, table);
  smmc_sim sim = make_sim(mode, seed, 0, static_cast<std::uint64_t>(n_total), periods, capital);
  sim.gauss_mean = mean;
  sim.gauss_std = stddev;
  Progress prog{&n_simulations};
  check(smmc_group_set_progress(gs.group, &Progress::on_progress, &prog));
  const int rc = smmc_group_simulate(gs.group, &sim, out, nullptr, nullptr, nullptr, nullptr, nullptr);
  (void)smmc_group_set_progress(gs.group, nullptr, nullptr);
  check(rc);
  n_simulations = n_total;  // src/simulations.cu:678
}

// Sizes `v` to n floats with its pages already resident.  A fresh 400 MB vector costs one page
// fault per 4 KiB when resize() zero-fills it (60-120 ms measured on the GPU box's host: more than
// the whole simulation).  So: reserve, have the kernel map the new allocation (madvise
// MADV_POPULATE_WRITE: no user-space write into unconstructed storage), then resize -- the
// value-initialisation then runs at memset speed.  Measured for 400 MB (tools/ubench_hostreg.cpp):
// plain resize 60 ms; populate from 1 / 2 / 4 / 8 threads + resize 28 / 17 / 24 / 20 ms: two threads.
// Any failure of the hint is ignored.
void resize_prefaulted_impl(std::vector<float> &v, size_t n) {
  v.clear();  // the result replaces the contents (src/simulations.cu:643-644): nothing to carry over
  if (n > v.capacity() && n * sizeof(float) >= (size_t(8) << 20)) {
    v.reserve(n);
    const long page = sysconf(_SC_PAGESIZE);
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(v.data()) + page - 1) / page * page;
    const uintptr_t hi = reinterpret_cast<uintptr_t>(v.data() + n) / page * page;
    if (page > 0 && hi > lo) {
      const uintptr_t mid = lo + (hi - lo) / 2 / page * page;
      std::thread upper([mid, hi] { (void)madvise(reinterpret_cast<void *>(mid), hi - mid, MADV_POPULATE_WRITE); });
      if (mid > lo) (void)madvise(reinterpret_cast<void *>(lo), mid - lo, MADV_POPULATE_WRITE);
      upper.join();
    }
  }
  v.resize(n);
}

// The callee-sized result of mc_simulations_gpu (src/simulations.cu:643-644), made ready beside the
// engines.  In a fresh process the group's engines cost the HIP runtime's start-up (190-290 ms) plus
// 25-40 ms of their own; sizing the 400 MB vector costs 17-20 ms and page-locking it 12-15 ms.  The
// engines (and the staging buffers of the host pipeline) come up on a helper thread while this thread
// sizes the vector and registers it, so that the first smmc_group_simulate finds everything in place
// (profiles/r03/cold_start.txt has the phases).  The registration is released by PinnedResult.
struct PinnedResult {
  void *ptr = nullptr;
  // SMMC_PIN_HOST (smmc_host.h): only "whole" (the default) registers here, beside the engines' start-up.
  // "chunk" is the engine's own policy -- it registers chunk by chunk ahead of its copies -- and a buffer found
  // pinned would switch that off (the cold-start rows labelled "chunk" had measured "whole").
  void pin(std::vector<float> &v) {
    if (smmc::pin_policy_from_env() != smmc::kPinWhole || v.size() * sizeof(float) < smmc::kPinMinBytes) return;
    if (smmc_host_register(v.data(), v.size() * sizeof(float)) == SMMC_OK) ptr = v.data();  // failure: pageable copies
  }
  ~PinnedResult() {
    if (ptr) (void)smmc_host_unregister(ptr);
  }
};

void size_result_and_warm(std::vector<float> &totals, long n_total, int n_gpus, const std::vector<float> *table,
                          PinnedResult &pinned, int mode, unsigned periods, float capital) {
  const auto t0 = std::chrono::steady_clock::now();
  std::exception_ptr warm_error;
  double t_warm = 0.0;
  std::thread warm([&] {
    try {
      GroupSession gs(n_gpus, table);
      check(smmc_group_prepare_host(gs.group, static_cast<std::uint64_t>(n_total)));
      // a throw-away run of a few paths through the same kernel variant and the same pipeline: the
      // first launch and the first device-to-host copy of a process cost ~7 ms that the real run
      // would otherwise pay (its values go nowhere; the seed is the caller's business, not this one's)
      const std::uint64_t n_warm = static_cast<std::uint64_t>(std::min<long>(n_total, 256L * n_gpus));
      if (n_warm) {
        std::vector<float> scratch(n_warm);
        smmc_sim sim = make_sim(mode, 1, 0, n_warm, periods, capital);
        sim.flags |= SMMC_FLAG_QUIET;
        check(smmc_group_simulate(gs.group, &sim, scratch.data(), nullptr, nullptr, nullptr, nullptr, nullptr));
      }
    } catch (...) {
      warm_error = std::current_exception();
    }
    t_warm = seconds_since(t0);
  });
  double t_sized = 0.0, t_pinned = 0.0;
  try {
    resize_prefaulted_impl(totals, static_cast<size_t>(n_total));
    t_sized = seconds_since(t0);
    pinned.pin(totals);
    t_pinned = seconds_since(t0);
  } catch (...) {  // e.g. std::bad_alloc for the result: the helper thread must be joined before the stack unwinds
    warm.join();
    throw;
  }
  warm.join();
  if (warm_error) std::rethrow_exception(warm_error);
  if (verbose())
    std::fprintf(stderr, "smmc: engines up in %.3f s; beside them: result vector (%ld floats) sized in %.3f s, page-locked in %.3f s\n",
                 t_warm, n_total, t_sized, t_pinned - t_sized);
}

}  // namespace

// ---- compounding core --------------------------------------------------------------------

float update_fund(float fund_value, float period_return) { return smmc_update_fund(fund_value, period_return); }

void __many_updates(float *returns, float *totals, unsigned int n_periods) {
  smmc_many_updates(returns, totals, n_periods);
}

std::vector<float> many_updates(float fund_value, std::vector<float> &returns, unsigned int n_periods) {
  if (returns.size() < n_periods) throw std::out_of_range("smmc: many_updates needs n_periods returns");
  std::vector<float> totals(static_cast<size_t>(n_periods) + 1);  // heap, not the reference's VLA (:28)
  totals[0] = fund_value;
  smmc_many_updates(returns.data(), totals.data(), n_periods);
  return totals;
}

std::vector<float> many_updates(float fund_value, std::vector<float> &returns, long n_updates) {
  if (n_updates < 0 || n_updates > 0xFFFFFFFFl) throw std::invalid_argument("smmc: n_updates out of range");
  return many_updates(fund_value, returns, static_cast<unsigned int>(n_updates));
}

// ---- returns sources -----------------------------------------------------------------------

std::vector<float> sample_returns_gaussian(unsigned int n, float return_mean, float return_std) {
  // src/simulations.cpp:41-55 (which writes into a reserve()d vector; here it is sized)
  std::mt19937_64 engine(next_seed());
  std::normal_distribution<float> dist(return_mean, return_std);
  std::vector<float> out(n);
  for (auto &v : out) v = dist(engine);
  return out;
}

std::vector<float> sample_returns_historical(unsigned int n, std::vector<float> &historical_returns) {
  // src/simulations.cpp:95-112
  if (historical_returns.empty()) throw std::out_of_range("smmc: empty returns table");
  std::mt19937 engine(static_cast<std::uint32_t>(next_seed()));
  std::uniform_int_distribution<int> pick(0, static_cast<int>(historical_returns.size()) - 1);
  std::vector<float> out;
  out.reserve(n);
  for (unsigned int i = 0; i < n; ++i) out.push_back(historical_returns.at(pick(engine)));
  return out;
}

std::vector<float> read_historical_returns(std::string csv_fpath) {
  // src/simulations.cpp:83-93: column "returns" of a CSV; other columns ignored.  The file
  // python/get_data.py:59-69 writes has an empty cell in its first data row: skipped.
  std::ifstream in(csv_fpath);
  if (!in) throw std::runtime_error("smmc: cannot open " + csv_fpath);
  std::string line;
  if (!std::getline(in, line)) throw std::runtime_error("smmc: " + csv_fpath + " is empty");
  auto split = [](const std::string &l) {
    std::vector<std::string> cells;
    std::string cell;
    std::stringstream ss(l);
    while (std::getline(ss, cell, ',')) cells.push_back(cell);
    if (!l.empty() && l.back() == ',') cells.emplace_back();
    return cells;
  };
  auto trim = [](std::string s) {
    const char *ws = " \t\r\n\"";
    const size_t a = s.find_first_not_of(ws);
    if (a == std::string::npos) return std::string();
    return s.substr(a, s.find_last_not_of(ws) - a + 1);
  };
  int col = -1;
  const auto header = split(line);
  for (size_t i = 0; i < header.size(); ++i)
    if (trim(header[i]) == "returns") col = static_cast<int>(i);
  if (col < 0) throw std::runtime_error("smmc: " + csv_fpath + " has no column named returns");
  std::vector<float> out;
  while (std::getline(in, line)) {
    const auto cells = split(line);
    if (static_cast<size_t>(col) >= cells.size()) continue;
    const std::string cell = trim(cells[col]);
    if (cell.empty() || cell == "nan" || cell == "NaN") continue;
    out.push_back(std::strtof(cell.c_str(), nullptr));
  }
  return out;
}

// ---- engines -----------------------------------------------------------------------------------

void mc_simulations(std::atomic<long> &n_simulations, long max_n_simulations, unsigned int n_periods,
                    float initial_capital, std::vector<float> &historical_returns,
                    std::vector<float> &final_values) {
  if (max_n_simulations < 0) throw std::invalid_argument("smmc: negative max_n_simulations");
  if (final_values.size() < static_cast<size_t>(max_n_simulations))
    throw std::length_error("smmc: final_values must be pre-sized to max_n_simulations (src/simulations.cpp:252)");
  run_final_values(n_simulations, max_n_simulations, n_periods, initial_capital, SMMC_MODE_TABLE,
                   &historical_returns, 0.f, 0.f, final_values.data(), 1);
}

void mc_simulations_keepdata(std::atomic<long> &n_simulations, long max_n_simulations, unsigned int n_periods,
                             float initial_capital, std::vector<float> &historical_returns,
                             std::vector<std::vector<float>> &mc_data, std::vector<float> &final_values) {
  if (max_n_simulations < 0) throw std::invalid_argument("smmc: negative max_n_simulations");
  const size_t n = static_cast<size_t>(max_n_simulations);
  if (final_values.size() < n || mc_data.size() < n)
    throw std::length_error("smmc: mc_data and final_values must be pre-sized (src/simulations.cpp:183-184)");
  n_simulations = 0;
  const size_t row = static_cast<size_t>(n_periods) + 1;
  // slices bound the flat host staging buffer (~256 MiB) before rows go to their vectors
  const size_t slice = std::max<size_t>(1, (size_t(1) << 26) / row);
  std::vector<float> flat(std::min(n, slice) * row);
  const std::uint64_t seed = next_seed();
  Session ses(0, &historical_returns);
  for (size_t first = 0; first < n; first += slice) {
    const size_t count = std::min(slice, n - first);
    smmc_sim sim = make_sim(SMMC_MODE_TABLE, seed, first, count, n_periods, initial_capital);
    check(smmc_engine_simulate_keepdata_to_host(ses.engine, &sim, flat.data(), final_values.data() + first));
    for (size_t i = 0; i < count; ++i) mc_data[first + i].assign(flat.begin() + i * row, flat.begin() + (i + 1) * row);
    n_simulations += static_cast<long>(count);  // src/simulations.cpp:190
  }
}

void mc_simulations_gpu(std::atomic<long> &n_simulations, long max_n_simulations, int n_periods,
                        float initial_capital, std::vector<float> &returns, std::vector<float> &totals,
                        int n_gpus) {
  if (max_n_simulations < 0 || n_periods < 0) throw std::invalid_argument("smmc: negative size");
  PinnedResult pinned;
  size_result_and_warm(totals, max_n_simulations, n_gpus, &returns, pinned, SMMC_MODE_TABLE, static_cast<unsigned>(n_periods), initial_capital);  // callee-owned result, src/simulations.cu:643-644
  run_final_values(n_simulations, max_n_simulations, static_cast<unsigned>(n_periods), initial_capital,
                   SMMC_MODE_TABLE, &returns, 0.f, 0.f, totals.data(), n_gpus);
}

void mc_simulations_gpu_reduceBlock(std::atomic<long> &n_simulations, long max_n_simulations, int n_periods,
                                    float initial_capital, std::vector<float> &returns,
                                    std::vector<float> &means, std::vector<float> &variances, int n_gpus) {
  if (n_gpus != 1)  // src/simulations.cu:693
    throw std::invalid_argument("mc_simulations_gpu_reduceBlock: only 1 GPU supported");
  if (max_n_simulations < 0 || n_periods < 0) throw std::invalid_argument("smmc: negative size");
  const size_t n_chunks = (static_cast<size_t>(max_n_simulations) + SMMC_CHUNK - 1) / SMMC_CHUNK;
  means.assign(n_chunks, 0.f);  // src/simulations.cu:429-432
  variances.assign(n_chunks, 0.f);
  n_simulations = 0;
  Session ses(0, &returns);
  smmc_sim sim = make_sim(SMMC_MODE_TABLE, next_seed(), 0, static_cast<std::uint64_t>(max_n_simulations),
                          static_cast<unsigned>(n_periods), initial_capital);
  check(smmc_engine_simulate_to_host(ses.engine, &sim, nullptr, means.data(), variances.data(), nullptr, nullptr, nullptr));
  n_simulations = max_n_simulations;  // src/simulations.cu:695
}

float reduce_mean_gpu(std::vector<float> &vec, long n) {
  if (n <= 0 || static_cast<size_t>(n) > vec.size()) throw std::invalid_argument("smmc: reduce_mean_gpu needs 0 < n <= vec.size()");
  Session ses(0, nullptr);
  float mean = 0.f;
  check(smmc_engine_reduce_mean_host(ses.engine, vec.data(), static_cast<std::uint64_t>(n), &mean, nullptr));
  return mean;
}

// ---- CSV writers (src/helpers.cpp) ------------------------------------------------------------------

// ---- include/stock_market_monte_carlo/gpu.h: the reference's vector-add demo ----------------------

void vector_add(float *out, float *a, float *b, int n) {  // src/gpu.cpp:7-15
  const std::clock_t start = std::clock();
  for (int i = 0; i < n; i++) out[i] = a[i] + b[i];
  std::printf("CPU time: %f\n", static_cast<double>(std::clock() - start) / CLOCKS_PER_SEC);
}

void vector_add_gpu(float *out, float *a, float *b, int n) {  // src/gpu.cu:17-47
  double seconds = 0.0;
  if (smmc_vector_add(out, a, b, n, &seconds) != SMMC_OK) throw std::runtime_error(smmc_last_error());
  std::printf("GPU time: %f\n", seconds);
}

void print_vector(std::vector<float> &v) {
  std::printf("v = [ ");
  for (float x : v) std::printf("%6.3f ", x);
  std::printf(" ]\n");
}

void write_vector_file(std::string fname, std::vector<float> &v) {
  std::ofstream out(fname);
  for (float x : v) out << x << ",";
}

void write_data_file(std::string fname, std::vector<float> &returns, std::vector<float> &values) {
  std::printf("Writing data to csv file outputs/%s\n", fname.c_str());
  const std::string dir = "./outputs/";
  std::filesystem::create_directory(dir);
  std::ofstream out(dir + "/" + fname);
  out << "Returns,,";  // the double comma is the reference's (src/helpers.cpp:31): values[0] has no return
  for (float x : returns) out << x << ",";
  out << "\nValues,";
  for (float x : values) out << x << ",";
}

// ---- additions -----------------------------------------------------------------------------------

namespace smmc {

namespace {
void summary_of(const std::vector<float> &v, long n_el, float below, smmc_stats *st, float *quart) {
  if (n_el <= 0 || static_cast<size_t>(n_el) > v.size()) throw std::invalid_argument("smmc: need 0 < n_el <= vec.size()");
  Session ses(0, nullptr);
  check(smmc_engine_host_values_summary(ses.engine, v.data(), static_cast<std::uint64_t>(n_el), below, 0, 0.f, 1.f, st,
                                        nullptr, quart));
}
}  // namespace

void update_quartiles(std::vector<float> &quartiles, std::vector<float> &vec, long n_el) {
  float q[5];
  summary_of(vec, n_el, 0.f, nullptr, q);
  quartiles.assign(q, q + 5);  // {min, Q1, Q2, Q3, max}, examples/visualize_returns_cpu_v2.cpp:110
}

void update_mean_std(float &mean, float &std, std::vector<float> &v, long n_el) {
  smmc_stats st;
  summary_of(v, n_el, 0.f, &st, nullptr);
  // examples/visualize_returns_cpu_v2.cpp:113-123 sums (v - mean)^2, which cannot go negative; from
  // one pass the variance is formed with the DOUBLE mean and clamped, and only then rounded (with the
  // float-rounded mean the error 2 * mean * ulp(mean) swamps a small variance: 1000.013 +- 0.05 gave 0.18)
  const double m = st.sum / static_cast<double>(n_el);
  mean = static_cast<float>(m);
  std = static_cast<float>(std::sqrt(std::max(st.sumsq / static_cast<double>(n_el) - m * m, 0.0)));
}

long update_count_below_min(float &min_final_amount, const std::vector<float> &final_values, long n_simulations) {
  smmc_stats st;
  summary_of(final_values, n_simulations, min_final_amount, &st, nullptr);
  return static_cast<long>(st.below);
}

void resize_prefaulted(std::vector<float> &v, std::size_t n) { resize_prefaulted_impl(v, n); }

void fix_seed(bool fixed, std::uint64_t seed) {
  std::lock_guard<std::mutex> lock(g_seed_mutex);
  g_seed_fixed = fixed;
  g_seed = seed;
}

void mc_simulations_gpu_gaussian(std::atomic<long> &n_simulations, long max_n_simulations, int n_periods,
                                 float initial_capital, float return_mean, float return_std,
                                 std::vector<float> &totals, int n_gpus) {
  if (max_n_simulations < 0 || n_periods < 0) throw std::invalid_argument("smmc: negative size");
  PinnedResult pinned;
  size_result_and_warm(totals, max_n_simulations, n_gpus, nullptr, pinned, SMMC_MODE_GAUSSIAN, static_cast<unsigned>(n_periods), initial_capital);
  run_final_values(n_simulations, max_n_simulations, static_cast<unsigned>(n_periods), initial_capital,
                   SMMC_MODE_GAUSSIAN, nullptr, return_mean, return_std, totals.data(), n_gpus);
}

double Summary::stddev() const {
  if (!count) return 0.0;
  const double m = mean();
  return std::sqrt(std::max(sumsq / double(count) - m * m, 0.0));
}

Summary mc_summary(long max_n_simulations, int n_periods, float initial_capital, bool gaussian,
                   std::vector<float> &returns, float return_mean, float return_std, float below_threshold,
                   unsigned n_bins, float hist_lo, float hist_hi, int n_gpus) {
  if (max_n_simulations < 0 || n_periods < 0) throw std::invalid_argument("smmc: negative size");
  const std::uint64_t seed = next_seed();
  const size_t rec = static_cast<size_t>(smmc_stats_bytes(n_bins));
  std::vector<char> record(rec);
  {
    GroupSession gs(n_gpus, gaussian ? nullptr : &returns);
    smmc_sim sim = make_sim(gaussian ? SMMC_MODE_GAUSSIAN : SMMC_MODE_TABLE, seed, 0, static_cast<std::uint64_t>(max_n_simulations),
                            static_cast<unsigned>(n_periods), initial_capital);
    sim.gauss_mean = return_mean;
    sim.gauss_std = return_std;
    sim.below_threshold = below_threshold;
    sim.n_bins = n_bins;
    sim.hist_lo = hist_lo;
    sim.hist_hi = hist_hi;
    check(smmc_group_simulate(gs.group, &sim, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<smmc_stats *>(record.data()),
                              reinterpret_cast<uint64_t *>(record.data() + sizeof(smmc_stats))));
  }
  const smmc_stats *h = reinterpret_cast<const smmc_stats *>(record.data());
  Summary out;
  out.count = h->count;
  out.below = h->below;
  out.underflow = h->underflow;
  out.overflow = h->overflow;
  out.sum = h->sum;
  out.sumsq = h->sumsq;
  out.min = h->min;
  out.max = h->max;
  const uint64_t *bins = reinterpret_cast<const uint64_t *>(h + 1);
  out.hist.assign(bins, bins + n_bins);
  return out;
}

std::vector<float> bundled_synthetic_returns() {
  static const float table[] = {
#include "smmc_synthetic_table.inc"
  };
  return std::vector<float>(table, table + sizeof(table) / sizeof(table[0]));
}

}  // namespace smmc
