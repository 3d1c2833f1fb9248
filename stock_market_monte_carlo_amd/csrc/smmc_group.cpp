// smmc_group.cpp -- several devices of one process behind one call (the smmc_group_* entries of smmc.h).
//
// Counterpart of mc_simulations_multi_gpu_launcher_async (reference src/simulations.cu:576-655): shard the
// request over the devices, run all of them at once, put every shard's final values at its offset of the
// caller's array, and return ONE statistics record.  One host thread per device drives that device's
// engine (smmc_capi.cpp); the engines know nothing of each other.  The record merge has two back-ends:
// a host loop over the 864-byte records in device order, and RCCL -- ncclCommInitAll once per group,
// one grouped all-reduce of the integer fields per call.  librccl (0.3-0.6 GB on disk) is opened with
// dlopen only when a group asks for it: the library itself does not link it, so programs that never
// use it do not pay for loading it.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // types and prototypes only; the functions come from dlsym

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "smmc.h"
#include "smmc_host.h"

namespace {

double ms_since(const std::chrono::steady_clock::time_point &t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

bool verbose_env() {
  const char *env = std::getenv("SMMC_VERBOSE");
  return env && *env && *env != '0';
}

// The RCCL entry points a group uses, resolved once per process.
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      const char *why = dlerror();
      r.error = std::string("librccl could not be opened: ") + (why ? why : "?");
      return;
    }
    auto sym = [&](const char *name) {
      void *p = dlsym(r.lib, name);
      if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + name;
      return p;
    };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return r;
}

}  // namespace

// smmc_capi.cpp: stores a message for smmc_last_error() of the calling thread
extern "C" __attribute__((visibility("hidden"))) int smmc_set_error_(int code, const char *message);

namespace {

int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  return smmc_set_error_(code, buf);
}

}  // namespace

struct smmc_group {
  int merge = SMMC_MERGE_HOST;
  std::vector<int> devices;
  std::vector<smmc_engine *> engines;
  std::vector<hipStream_t> streams;  // the engines' own streams
  std::vector<void *> d_record;      // per device: one packed record with SMMC_MAX_BINS buckets (RCCL merge)
  std::vector<ncclComm_t> comms;
  smmc_progress_fn progress_fn = nullptr;
  void *progress_user = nullptr;
  double engines_ms = 0.0, comm_init_ms = 0.0, last_merge_ms = 0.0;
  // the table every device holds (smmc_group_set_table): an identical one is not uploaded again, and a
  // set_table that failed on some device leaves the group without a valid table until one succeeds everywhere
  std::vector<float> table;
  bool table_valid = true;  // false only between a failed smmc_group_set_table and the next successful one
};

namespace {

struct ShardProgress {  // one per device and call: deltas go to the call's total
  std::atomic<int64_t> *total;
  volatile int64_t *user_counter;
  smmc_group *group;
  std::mutex *report_lock;
  int64_t reported = 0;
  static void on_progress(void *self, int64_t finished) {
    ShardProgress *p = static_cast<ShardProgress *>(self);
    if (finished <= p->reported) return;
    // the total is advanced UNDER the lock: two devices that finish a chunk at once must report in the order of their
    // totals (found with eight fake devices: 7 then 5), and callers' callbacks are not asked to be re-entrant
    std::lock_guard<std::mutex> lock(*p->report_lock);
    const int64_t now = p->total->fetch_add(finished - p->reported, std::memory_order_acq_rel) + (finished - p->reported);
    p->reported = finished;
    if (p->user_counter) {
      int64_t cur = __atomic_load_n(const_cast<int64_t *>(p->user_counter), __ATOMIC_RELAXED);
      if (now > cur) __atomic_store_n(const_cast<int64_t *>(p->user_counter), now, __ATOMIC_RELEASE);
    }
    if (p->group->progress_fn) p->group->progress_fn(p->group->progress_user, now);
  }
};

// fn(i) for i = 0 .. n - 1, one host thread each (the calling thread alone when n == 1).  False if a thread
// could not be started: those that were are joined first, the others' work was not done.
template <typename Fn>
bool run_on_threads(int n, Fn fn) {
  if (n == 1) {
    fn(0);
    return true;
  }
  std::vector<std::thread> threads;
  bool ok = true;
  try {
    for (int i = 0; i < n; ++i) threads.emplace_back(fn, i);
  } catch (...) {
    ok = false;
  }
  for (auto &t : threads) t.join();
  return ok;
}

void shard_of(uint64_t n, int n_dev, int index, uint64_t *first, uint64_t *count) {
  const uint64_t base = n / n_dev, extra = n % n_dev, i = static_cast<uint64_t>(index);
  *first = base * i + (i < extra ? i : extra);
  *count = base + (i < extra ? 1 : 0);
}

}  // namespace

extern "C" {

int smmc_group_create(const int *devices, int n_devices, int merge, smmc_group **out) {
  if (!out) return fail(SMMC_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return fail(SMMC_ERR_INVALID, "a group has 1 .. 64 devices");
  if (merge != SMMC_MERGE_HOST && merge != SMMC_MERGE_RCCL) return fail(SMMC_ERR_INVALID, "unknown merge back-end %d", merge);
  if (merge == SMMC_MERGE_RCCL)
    for (int a = 0; a < n_devices; ++a)
      for (int b = a + 1; b < n_devices; ++b)
        if (devices[a] == devices[b])
          return fail(SMMC_ERR_INVALID, "SMMC_MERGE_RCCL needs distinct devices (device %d is listed twice); use SMMC_MERGE_HOST",
                      devices[a]);
  smmc_group *g = new (std::nothrow) smmc_group();
  if (!g) return fail(SMMC_ERR_NOMEM, "out of host memory");
  g->merge = merge;
  g->devices.assign(devices, devices + n_devices);
  g->engines.assign(n_devices, nullptr);
  g->streams.assign(n_devices, nullptr);
  g->d_record.assign(n_devices, nullptr);
  const auto t0 = std::chrono::steady_clock::now();
  // engines in parallel: in a fresh process the first one pays the HIP runtime's start-up, the others
  // their device's context
  std::vector<int> rcs(n_devices, SMMC_OK);
  std::vector<std::string> errs(n_devices);
  auto make = [&](int i) {
    rcs[i] = smmc_engine_create(g->devices[i], SMMC_STREAM_NEW, &g->engines[i]);
    if (rcs[i] == SMMC_OK) {
      void *s = nullptr;
      rcs[i] = smmc_engine_get_stream(g->engines[i], &s);
      g->streams[i] = static_cast<hipStream_t>(s);
    }
    if (rcs[i] != SMMC_OK) errs[i] = smmc_last_error();
  };
  if (!run_on_threads(n_devices, make)) {
    smmc_group_destroy(g);
    return fail(SMMC_ERR_NOMEM, "could not start the host threads that create the engines");
  }
  g->engines_ms = ms_since(t0);
  for (int i = 0; i < n_devices; ++i)
    if (rcs[i] != SMMC_OK) {
      const int rc = rcs[i];
      const std::string msg = errs[i];
      smmc_group_destroy(g);
      return fail(rc, "device %d: %s", devices[i], msg.c_str());
    }
  if (merge == SMMC_MERGE_RCCL) {
    const auto t1 = std::chrono::steady_clock::now();  // comm_init_ms: opening librccl (first group of the process) + ncclCommInitAll
    Rccl &r = rccl();
    if (!r.error.empty()) {
      const std::string msg = r.error;
      smmc_group_destroy(g);
      return fail(SMMC_ERR_INVALID, "%s", msg.c_str());
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int i = 0; i < n_devices; ++i) {
      hipError_t err = hipSetDevice(g->devices[i]);
      if (err == hipSuccess) err = hipMalloc(&g->d_record[i], smmc_stats_bytes(SMMC_MAX_BINS));
      if (err != hipSuccess) {
        if (prev >= 0) (void)hipSetDevice(prev);
        smmc_group_destroy(g);
        return fail(SMMC_ERR_HIP, "hipMalloc of the record on device %d failed: %s", devices[i], hipGetErrorString(err));
      }
    }
    g->comms.assign(n_devices, nullptr);
    const ncclResult_t res = r.CommInitAll(g->comms.data(), n_devices, g->devices.data());
    g->comm_init_ms = ms_since(t1);
    if (prev >= 0) (void)hipSetDevice(prev);
    if (res != ncclSuccess) {
      g->comms.clear();
      smmc_group_destroy(g);
      return fail(SMMC_ERR_HIP, "ncclCommInitAll over %d device(s) failed: %s", n_devices, r.GetErrorString(res));
    }
  }
  if (verbose_env())
    std::fprintf(stderr, "smmc: group of %d device(s): engines up in %.3f s%s\n", n_devices, g->engines_ms / 1e3,
                 merge == SMMC_MERGE_RCCL ? (", RCCL communicator in " + std::to_string(g->comm_init_ms / 1e3) + " s").c_str() : "");
  *out = g;
  return SMMC_OK;
}

void smmc_group_destroy(smmc_group *g) {
  if (!g) return;
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (!g->comms.empty()) {
    Rccl &r = rccl();
    for (ncclComm_t c : g->comms)
      if (c) (void)r.CommDestroy(c);
  }
  for (size_t i = 0; i < g->engines.size(); ++i) {
    if (g->d_record[i] && hipSetDevice(g->devices[i]) == hipSuccess) (void)hipFree(g->d_record[i]);
    if (g->engines[i]) smmc_engine_destroy(g->engines[i]);
  }
  if (prev >= 0) (void)hipSetDevice(prev);
  delete g;
}

int smmc_group_size(const smmc_group *g) { return g ? static_cast<int>(g->devices.size()) : 0; }

int smmc_group_set_table(smmc_group *g, const float *returns_percent, uint32_t n) {
  if (!g) return fail(SMMC_ERR_INVALID, "group is NULL");
  if (!returns_percent || n == 0) return fail(SMMC_ERR_INVALID, "empty returns table");
  // the same bytes as every device already holds: nothing to do (the drop-in sets the table before every call,
  // twice per mc_simulations_gpu: for the warm-up run and for the run)
  if (g->table_valid && g->table.size() == n && !std::memcmp(g->table.data(), returns_percent, sizeof(float) * n)) return SMMC_OK;
  // From the first upload on the devices may disagree: a failure half way leaves the earlier devices with the
  // new table and the later ones with the old.  The group then has NO valid table -- smmc_group_simulate
  // refuses table mode -- until a set_table succeeds on every device.
  g->table_valid = false;
  g->table.clear();
  for (size_t i = 0; i < g->engines.size(); ++i) {
    const int rc = smmc_engine_set_table(g->engines[i], returns_percent, n);
    if (rc != SMMC_OK) {
      const std::string msg = smmc_last_error();
      return fail(rc, "device %d: %s (the group's devices now hold different tables: set the table again)", g->devices[i], msg.c_str());
    }
  }
  g->table.assign(returns_percent, returns_percent + n);
  g->table_valid = true;
  return SMMC_OK;
}

int smmc_group_set_progress(smmc_group *g, smmc_progress_fn fn, void *user) {
  if (!g) return fail(SMMC_ERR_INVALID, "group is NULL");
  g->progress_fn = fn;
  g->progress_user = fn ? user : nullptr;
  return SMMC_OK;
}

int smmc_group_prepare_host(smmc_group *g, uint64_t n_paths) {
  if (!g) return fail(SMMC_ERR_INVALID, "group is NULL");
  const int G = static_cast<int>(g->devices.size());
  std::vector<int> rcs(G, SMMC_OK);
  std::vector<std::string> errs(G);
  auto run = [&](int i) {
    uint64_t first, count;
    shard_of(n_paths, G, i, &first, &count);
    rcs[i] = smmc_engine_prepare_host(g->engines[i], count);
    if (rcs[i] != SMMC_OK) errs[i] = smmc_last_error();
  };
  if (!run_on_threads(G, run)) return fail(SMMC_ERR_NOMEM, "could not start a host thread per device");
  for (int i = 0; i < G; ++i)
    if (rcs[i] != SMMC_OK) return fail(rcs[i], "device %d: %s", g->devices[i], errs[i].c_str());
  return SMMC_OK;
}

int smmc_group_shard(const smmc_group *g, uint64_t n_paths, int index, uint64_t *first, uint64_t *count) {
  if (!g || !first || !count) return fail(SMMC_ERR_INVALID, "NULL argument");
  if (index < 0 || index >= static_cast<int>(g->devices.size())) return fail(SMMC_ERR_INVALID, "no device %d in the group", index);
  shard_of(n_paths, static_cast<int>(g->devices.size()), index, first, count);
  return SMMC_OK;
}

int smmc_group_device_record(smmc_group *g, int index, void **d_record) {
  if (!g || !d_record) return fail(SMMC_ERR_INVALID, "NULL argument");
  if (g->merge != SMMC_MERGE_RCCL) return fail(SMMC_ERR_INVALID, "only a SMMC_MERGE_RCCL group keeps the merged record on its devices");
  if (index < 0 || index >= static_cast<int>(g->devices.size())) return fail(SMMC_ERR_INVALID, "no device %d in the group", index);
  *d_record = g->d_record[index];
  return SMMC_OK;
}

int smmc_group_timings(const smmc_group *g, double *engines_ms, double *comm_init_ms, double *last_merge_ms) {
  if (!g) return fail(SMMC_ERR_INVALID, "group is NULL");
  if (engines_ms) *engines_ms = g->engines_ms;
  if (comm_init_ms) *comm_init_ms = g->comm_init_ms;
  if (last_merge_ms) *last_merge_ms = g->last_merge_ms;
  return SMMC_OK;
}

int smmc_group_simulate(smmc_group *g, const smmc_sim *sim, float *host_final, float *host_chunk_mean,
                        float *host_chunk_var, volatile int64_t *progress, smmc_stats *stats, uint64_t *hist) {
  if (!g) return fail(SMMC_ERR_INVALID, "group is NULL");
  if (!sim) return fail(SMMC_ERR_INVALID, "sim is NULL");
  if (sim->struct_size != sizeof(smmc_sim))
    return fail(SMMC_ERR_INVALID, "smmc_sim.struct_size is %u, this library expects %zu", sim->struct_size, sizeof(smmc_sim));
  if (sim->n_bins > SMMC_MAX_BINS) return fail(SMMC_ERR_INVALID, "n_bins %u exceeds SMMC_MAX_BINS %d", sim->n_bins, SMMC_MAX_BINS);
  if (sim->mode == SMMC_MODE_TABLE && !g->table_valid)
    return fail(SMMC_ERR_INVALID, "the last smmc_group_set_table failed on some device: the devices hold different tables; set the table again");
  const int G = static_cast<int>(g->devices.size());
  const uint64_t n = sim->n_paths;
  const bool want_stats = stats != nullptr || hist != nullptr;
  const bool want_cs = host_chunk_mean != nullptr || host_chunk_var != nullptr;
  if (want_cs)
    for (int i = 0; i < G; ++i) {
      uint64_t first, count;
      shard_of(n, G, i, &first, &count);
      if (count && first % SMMC_CHUNK)
        return fail(SMMC_ERR_INVALID, "chunk means need every shard to start on a multiple of %d paths (shard %d starts at %llu)",
                    SMMC_CHUNK, i, static_cast<unsigned long long>(first));
    }
  const size_t rec = static_cast<size_t>(smmc_stats_bytes(sim->n_bins));
  std::vector<std::vector<char>> records(G, std::vector<char>(rec, 0));

  // One registration of the caller's result for all devices (each engine would otherwise register its
  // own share, and neighbouring shares overlap in the page their boundary falls in: ADVICE r2).  The rules are
  // the engine's (smmc_host.h): policy, threshold, both ends tested for "pinned already".  With several
  // devices "chunk" means "whole" here (chunk-wise ownership of pages is an engine's own business), and the
  // shards are told not to register anything themselves -- also when this registration fails (the
  // engines would then each try their own share and collide in the boundary pages).
  void *pinned = nullptr;
  uint32_t shard_flags = 0;
  if (host_final && G > 1) {
    shard_flags = SMMC_FLAG_HOST_NOPIN;
    const uint64_t bytes = sizeof(float) * n;
    if (bytes >= smmc::kPinMinBytes && smmc::pin_policy_from_env() != smmc::kPinNever && !smmc::host_range_is_pinned(host_final, bytes)) {
      const uintptr_t page = 4096, lo = reinterpret_cast<uintptr_t>(host_final) & ~(page - 1),
                      hi = (reinterpret_cast<uintptr_t>(host_final + n) + page - 1) & ~(page - 1);
      if (hipHostRegister(reinterpret_cast<void *>(lo), hi - lo, hipHostRegisterPortable) == hipSuccess) {
        pinned = reinterpret_cast<void *>(lo);
      } else {
        if (verbose_env())
          std::fprintf(stderr, "smmc: hipHostRegister of the %zu-byte result failed (%s): pageable copies\n",
                       static_cast<size_t>(hi - lo), hipGetErrorString(hipGetLastError()));
        else
          (void)hipGetLastError();
      }
    }
  }

  std::atomic<int64_t> total{0};
  std::mutex report_lock;
  std::vector<ShardProgress> prog(G);
  const bool polled = progress != nullptr || g->progress_fn != nullptr;
  if (progress) __atomic_store_n(const_cast<int64_t *>(progress), static_cast<int64_t>(0), __ATOMIC_RELEASE);
  if (g->progress_fn) g->progress_fn(g->progress_user, 0);
  std::vector<int> rcs(G, SMMC_OK);
  std::vector<std::string> errs(G);
  const auto t_all = std::chrono::steady_clock::now();
  auto run = [&](int i) {
    uint64_t first, count;
    shard_of(n, G, i, &first, &count);
    const auto t0 = std::chrono::steady_clock::now();
    smmc_sim part = *sim;
    part.first_path = sim->first_path + first;
    part.n_paths = count;
    part.flags |= shard_flags;
    smmc_engine *e = g->engines[i];
    if (polled) {
      prog[i] = ShardProgress{&total, progress, g, &report_lock, 0};
      (void)smmc_engine_set_progress(e, &ShardProgress::on_progress, &prog[i]);
    }
    smmc_stats *hdr = reinterpret_cast<smmc_stats *>(records[i].data());
    rcs[i] = smmc_engine_simulate_to_host(e, &part, host_final ? host_final + first : nullptr,
                                          host_chunk_mean ? host_chunk_mean + first / SMMC_CHUNK : nullptr,
                                          host_chunk_var ? host_chunk_var + first / SMMC_CHUNK : nullptr, nullptr,
                                          want_stats ? hdr : nullptr,
                                          want_stats ? reinterpret_cast<uint64_t *>(records[i].data() + sizeof(smmc_stats)) : nullptr);
    if (polled) (void)smmc_engine_set_progress(e, nullptr, nullptr);
    if (rcs[i] != SMMC_OK) errs[i] = smmc_last_error();
    hdr->n_bins = sim->n_bins;
    if (verbose_env() && !(sim->flags & SMMC_FLAG_QUIET))  // phase timers, as the reference's launchers print them (src/simulations.cu:351-358,608-610)
      std::fprintf(stderr, "smmc: shard %d on device %d: paths [%llu, %llu): simulate+copy %.3f s\n", i, g->devices[i],
                   static_cast<unsigned long long>(first), static_cast<unsigned long long>(first + count), ms_since(t0) / 1e3);
  };
  const bool started = run_on_threads(G, run);
  if (pinned) (void)hipHostUnregister(pinned);
  if (!started) return fail(SMMC_ERR_NOMEM, "could not start a host thread per device");
  for (int i = 0; i < G; ++i)
    if (rcs[i] != SMMC_OK) return fail(rcs[i], "shard %d on device %d: %s", i, g->devices[i], errs[i].c_str());
  if (verbose_env() && !(sim->flags & SMMC_FLAG_QUIET))
    std::fprintf(stderr, "smmc: %llu paths x %u periods on %d shard(s): %.3f s\n", static_cast<unsigned long long>(n),
                 sim->n_periods, G, ms_since(t_all) / 1e3);

  if (want_stats) {
    const auto t_merge = std::chrono::steady_clock::now();
    std::vector<char> acc(records[0]);
    smmc_stats *h = reinterpret_cast<smmc_stats *>(acc.data());
    if (g->merge == SMMC_MERGE_HOST) {
      for (int i = 1; i < G; ++i) {  // device order
        const int rc = smmc_stats_merge(acc.data(), records[i].data());
        if (rc != SMMC_OK) return rc;
      }
    } else {
      // integers: ONE grouped all-reduce (header counters + bucket counts) after which every device
      // holds the merged integer record; doubles and min / max: host, device order
      Rccl &r = rccl();
      int prev = -1;
      (void)hipGetDevice(&prev);
      hipError_t err = hipSuccess;
      for (int i = 0; i < G && err == hipSuccess; ++i) {
        err = hipSetDevice(g->devices[i]);
        if (err == hipSuccess) err = hipMemcpyAsync(g->d_record[i], records[i].data(), rec, hipMemcpyHostToDevice, g->streams[i]);
      }
      ncclResult_t res = ncclSuccess;
      if (err == hipSuccess) {
        res = r.GroupStart();
        for (int i = 0; i < G && res == ncclSuccess; ++i) {
          char *d = static_cast<char *>(g->d_record[i]);
          res = r.AllReduce(d, d, 4, ncclUint64, ncclSum, g->comms[i], g->streams[i]);  // count, below, underflow, overflow
          if (res == ncclSuccess && sim->n_bins)
            res = r.AllReduce(d + sizeof(smmc_stats), d + sizeof(smmc_stats), sim->n_bins, ncclUint64, ncclSum, g->comms[i],
                              g->streams[i]);
        }
        const ncclResult_t end = r.GroupEnd();
        if (res == ncclSuccess) res = end;
      }
      std::vector<char> merged(rec);
      if (err == hipSuccess && res == ncclSuccess) {
        err = hipSetDevice(g->devices[0]);
        if (err == hipSuccess) err = hipMemcpyAsync(merged.data(), g->d_record[0], rec, hipMemcpyDeviceToHost, g->streams[0]);
        for (int i = 0; i < G && err == hipSuccess; ++i) {
          err = hipSetDevice(g->devices[i]);
          if (err == hipSuccess) err = hipStreamSynchronize(g->streams[i]);
        }
      }
      if (err != hipSuccess || res != ncclSuccess) {
        // copies from records[i] and into `merged` (pageable vectors) may still be queued: nothing may
        // touch them once this function has returned -- let every stream drain, errors ignored
        for (int i = 0; i < G; ++i)
          if (hipSetDevice(g->devices[i]) == hipSuccess) (void)hipStreamSynchronize(g->streams[i]);
        (void)hipGetLastError();
      }
      if (prev >= 0) (void)hipSetDevice(prev);
      if (res != ncclSuccess) return fail(SMMC_ERR_HIP, "RCCL all-reduce of the statistics record failed: %s", r.GetErrorString(res));
      if (err != hipSuccess) return fail(SMMC_ERR_HIP, "RCCL merge of the statistics record failed: %s", hipGetErrorString(err));
      const smmc_stats *m = reinterpret_cast<const smmc_stats *>(merged.data());
      h->count = m->count;
      h->below = m->below;
      h->underflow = m->underflow;
      h->overflow = m->overflow;
      std::memcpy(acc.data() + sizeof(smmc_stats), merged.data() + sizeof(smmc_stats), sizeof(uint64_t) * sim->n_bins);
      for (int i = 1; i < G; ++i) {  // device order, as smmc_stats_merge adds them
        const smmc_stats *s = reinterpret_cast<const smmc_stats *>(records[i].data());
        h->sum += s->sum;
        h->sumsq += s->sumsq;
        h->min = s->min < h->min ? s->min : h->min;
        h->max = s->max > h->max ? s->max : h->max;
      }
    }
    g->last_merge_ms = ms_since(t_merge);
    if (stats) *stats = *h;
    if (hist && sim->n_bins) std::memcpy(hist, acc.data() + sizeof(smmc_stats), sizeof(uint64_t) * sim->n_bins);
  }
  if (progress) __atomic_store_n(const_cast<int64_t *>(progress), static_cast<int64_t>(n), __ATOMIC_RELEASE);
  if (g->progress_fn) g->progress_fn(g->progress_user, static_cast<int64_t>(n));
  return SMMC_OK;
}

}  // extern "C"
