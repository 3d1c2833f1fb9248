// fake_rccl.cpp -- a HOST-MEMORY stand-in for librccl, for CPU tests of smmc_group's RCCL merge with SEVERAL devices
// (tests/test_group_fake_devices_cpu.py; TEST INFRASTRUCTURE, never linked into or loaded by the product outside
// those tests).
//
// csrc/smmc_group.cpp opens "librccl.so.1" with dlopen and calls ncclCommInitAll once per group and, per
// simulate call, ONE ncclGroupStart / ncclGroupEnd bracket holding two ncclAllReduce calls per device (the four
// header counters, the bucket counts; uint64, sum, in place).  Built as oracle/_san/fake_rccl_*/librccl.so.1 and put
// first on LD_LIBRARY_PATH, this file is what that dlopen finds in the fake-device tests: the buffers are host
// memory (tests/cpp/fake_hip.cpp), streams do nothing, and a collective completes inside ncclGroupEnd.  It checks
// what real RCCL would only show as a hang or as garbage on hardware the pool does not have: that every rank of a
// communicator posted the same number of collectives with the same count, type and operation, inside one bracket,
// each on its own communicator; then it reduces in rank order and writes every rank's receive buffer.
// Hooks for the test: fake_rccl_collectives() (completed all-reduces), fake_rccl_brackets(), fake_rccl_live_comms(),
// fake_rccl_fail_after(k) (the k-th next ncclAllReduce returns ncclInternalError; -1: none).
#include <rccl/rccl.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace {

struct Clique;
struct Call {
  const void *send;
  void *recv;
  size_t count;
  ncclDataType_t type;
  ncclRedOp_t op;
};

}  // namespace

struct ncclComm {  // the opaque type of rccl.h
  Clique *clique;
  int rank, device;
  std::vector<Call> pending;
};

namespace {

struct Clique {
  std::vector<ncclComm *> comms;
  int live = 0;
};

std::mutex g_mutex;
thread_local int t_depth = 0;
thread_local std::vector<ncclComm *> t_touched;  // communicators with calls posted in the open bracket of this thread
long g_collectives = 0, g_brackets = 0, g_live = 0, g_fail_after = -1;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclUint64: case ncclInt64: case ncclFloat64: return 8;
    case ncclUint32: case ncclInt32: case ncclFloat32: return 4;
    case ncclUint8: case ncclInt8: return 1;
    default: return 0;
  }
}

template <typename T>
void reduce(std::vector<char> &acc, const void *src, size_t count, ncclRedOp_t op, bool first) {
  T *a = reinterpret_cast<T *>(acc.data());
  const T *s = static_cast<const T *>(src);
  for (size_t i = 0; i < count; ++i) {
    if (first) a[i] = s[i];
    else if (op == ncclSum) a[i] = static_cast<T>(a[i] + s[i]);
    else if (op == ncclMin) a[i] = s[i] < a[i] ? s[i] : a[i];
    else if (op == ncclMax) a[i] = s[i] > a[i] ? s[i] : a[i];
  }
}

// every communicator of the clique has posted: the k-th call of each rank is one collective
ncclResult_t complete(Clique *q) {
  const size_t n_calls = q->comms[0]->pending.size();
  for (ncclComm *c : q->comms)
    if (c->pending.size() != n_calls) return ncclInvalidUsage;  // a rank posted more or fewer collectives: a hang on hardware
  for (size_t k = 0; k < n_calls; ++k) {
    const Call &c0 = q->comms[0]->pending[k];
    const size_t bytes = c0.count * type_bytes(c0.type);
    if (!type_bytes(c0.type) || (c0.op != ncclSum && c0.op != ncclMin && c0.op != ncclMax)) return ncclInvalidArgument;
    std::vector<char> acc(bytes);
    for (size_t r = 0; r < q->comms.size(); ++r) {
      const Call &c = q->comms[r]->pending[k];
      if (c.count != c0.count || c.type != c0.type || c.op != c0.op) return ncclInvalidArgument;  // ranks disagree
      if (!c.send || !c.recv) return ncclInvalidArgument;
      switch (c.type) {
        case ncclUint64: reduce<uint64_t>(acc, c.send, c.count, c.op, r == 0); break;
        case ncclInt64: reduce<int64_t>(acc, c.send, c.count, c.op, r == 0); break;
        case ncclFloat64: reduce<double>(acc, c.send, c.count, c.op, r == 0); break;
        case ncclUint32: reduce<uint32_t>(acc, c.send, c.count, c.op, r == 0); break;
        case ncclInt32: reduce<int32_t>(acc, c.send, c.count, c.op, r == 0); break;
        case ncclFloat32: reduce<float>(acc, c.send, c.count, c.op, r == 0); break;
        default: return ncclInvalidArgument;
      }
    }
    for (ncclComm *c : q->comms) std::memcpy(c->pending[k].recv, acc.data(), bytes);  // in place or not: acc is a copy
    ++g_collectives;
  }
  for (ncclComm *c : q->comms) c->pending.clear();
  return ncclSuccess;
}

ncclResult_t flush() {  // the bracket of this thread closes
  ncclResult_t res = ncclSuccess;
  std::lock_guard<std::mutex> lock(g_mutex);
  ++g_brackets;
  std::vector<Clique *> seen;
  for (ncclComm *c : t_touched) {
    Clique *q = c->clique;
    bool dup = false;
    for (Clique *s : seen) dup = dup || s == q;
    if (dup) continue;
    seen.push_back(q);
    // a single-process clique: every rank must have posted in THIS bracket (another thread's bracket cannot complete it)
    for (ncclComm *m : q->comms) {
      bool here = false;
      for (ncclComm *t : t_touched) here = here || t == m;
      if (!here && res == ncclSuccess) res = ncclInvalidUsage;
    }
    if (res == ncclSuccess) res = complete(q);
    if (res != ncclSuccess)
      for (ncclComm *m : q->comms) m->pending.clear();
  }
  t_touched.clear();
  return res;
}

}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist) {
  if (!comm || ndev < 1) return ncclInvalidArgument;
  for (int i = 0; i < ndev; ++i)
    for (int j = 0; j < i; ++j)
      if (devlist && devlist[i] == devlist[j]) return ncclInvalidUsage;  // as RCCL: one rank per device in a process clique
  std::lock_guard<std::mutex> lock(g_mutex);
  Clique *q = new Clique;
  for (int i = 0; i < ndev; ++i) {
    ncclComm *c = new ncclComm{q, i, devlist ? devlist[i] : i, {}};
    q->comms.push_back(c);
    comm[i] = c;
  }
  q->live = ndev;
  g_live += ndev;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lock(g_mutex);
  Clique *q = comm->clique;
  --g_live;
  if (--q->live == 0) {
    for (ncclComm *c : q->comms) delete c;
    delete q;
  }
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++t_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  return flush();
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t) {
  if (!comm) return ncclInvalidArgument;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_fail_after == 0) {
      g_fail_after = -1;
      return ncclInternalError;
    }
    if (g_fail_after > 0) --g_fail_after;
    comm->pending.push_back(Call{sendbuff, recvbuff, count, datatype, op});
  }
  bool known = false;
  for (ncclComm *t : t_touched) known = known || t == comm;
  if (!known) t_touched.push_back(comm);
  if (t_depth == 0) return flush();  // outside a bracket: completes at once (only a one-rank clique can)
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument (fake RCCL: the ranks of a collective disagree)";
    case ncclInvalidUsage: return "invalid usage (fake RCCL: a rank is missing from the bracket, or posted a different number of collectives)";
    case ncclInternalError: return "internal error (fake RCCL: injected)";
    default: return "unhandled error";
  }
}

long fake_rccl_collectives(void) { std::lock_guard<std::mutex> lock(g_mutex); return g_collectives; }
long fake_rccl_brackets(void) { std::lock_guard<std::mutex> lock(g_mutex); return g_brackets; }
long fake_rccl_live_comms(void) { std::lock_guard<std::mutex> lock(g_mutex); return g_live; }
void fake_rccl_fail_after(long k) { std::lock_guard<std::mutex> lock(g_mutex); g_fail_after = k; }

}  // extern "C"
