// fake_hip.cpp -- a HOST-MEMORY stand-in for the HIP runtime, for CPU tests of the product's host-side code with
// SEVERAL devices (tests/test_group_fake_devices_cpu.py; TEST INFRASTRUCTURE, never linked into the product).
//
// Linked INSTEAD of libamdhip64 together with tests/cpp/launch_fake.cpp (which stands in for the kernel
// translation units and computes a known function of the global path id on the host): csrc/smmc_capi.cpp,
// smmc_group.cpp and smmc_dropin.cpp then run unchanged -- engines, streams, staging buffers, the chunked host
// pipeline, one host thread per device, the one-time registration of the caller's buffer, the record merge --
// on FAKE_HIP_DEVICES (default 3) "devices" whose memory is the host's.  Everything is synchronous: a copy is a
// memcpy at enqueue time, an event is complete when recorded.  What this gives the one-GPU pool is G > 1 under
// ThreadSanitizer / AddressSanitizer: the per-device threads, the progress callbacks, the
// shared registration and the merge really run with three distinct devices before any multi-GPU hardware does.
// hipHostRegister keeps a table of registered ranges and refuses overlaps like the real runtime, so that a
// double registration of a boundary page shows up here as it would on the device.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>

namespace {

int device_count() {
  static const int n = [] {
    const char *env = std::getenv("FAKE_HIP_DEVICES");
    const int v = env ? std::atoi(env) : 3;
    return v >= 0 && v <= 64 ? v : 3;
  }();
  return n;
}

thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;

hipError_t done(hipError_t e) {
  if (e != hipSuccess) t_last = e;
  return e;
}

// never destroyed: the product's own statics (cached engines and groups) are torn down at exit and still call in
struct State {
  std::mutex mutex;
  std::map<void *, int> allocs;                 // device allocation -> device
  std::map<uintptr_t, uintptr_t> registered;    // page-locked host ranges: begin -> end
  std::set<void *> streams, events;
  int fail_malloc_device = -1;                  // fake_hip_fail_mallocs_on(): every hipMalloc on this device fails
};
State &state() {
  static State *s = new State;
  return *s;
}
#define g_mutex state().mutex
#define g_allocs state().allocs
#define g_registered state().registered
#define g_streams state().streams
#define g_events state().events
#define g_fail_malloc_device state().fail_malloc_device

}  // namespace

// test hooks (declared in the tests that use them)
extern "C" void fake_hip_fail_mallocs_on(int device) {
  std::lock_guard<std::mutex> lock(g_mutex);
  g_fail_malloc_device = device;
}
extern "C" size_t fake_hip_live_allocations(void) {
  std::lock_guard<std::mutex> lock(g_mutex);
  return g_allocs.size();
}
extern "C" size_t fake_hip_registered_ranges(void) {
  std::lock_guard<std::mutex> lock(g_mutex);
  return g_registered.size();
}

extern "C" {

hipError_t hipGetDeviceCount(int *count) {
  *count = device_count();
  return done(device_count() ? hipSuccess : hipErrorNoDevice);
}
hipError_t hipSetDevice(int deviceId) {
  if (deviceId < 0 || deviceId >= device_count()) return done(hipErrorInvalidDevice);
  t_device = deviceId;
  return hipSuccess;
}
hipError_t hipGetDevice(int *deviceId) {
  *deviceId = t_device;
  return hipSuccess;
}
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int deviceId) {
  if (deviceId < 0 || deviceId >= device_count()) return done(hipErrorInvalidDevice);
  std::memset(prop, 0, sizeof *prop);
  std::strcpy(prop->gcnArchName, "gfx950:fake");
  std::strcpy(prop->name, "fake MI355X (host memory)");
  prop->multiProcessorCount = 4;
  prop->sharedMemPerBlock = 64 * 1024;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) {
  const hipError_t e = t_last;
  t_last = hipSuccess;
  return e;
}
const char *hipGetErrorString(hipError_t e) {
  switch (e) {
    case hipSuccess: return "no error";
    case hipErrorNoDevice: return "no device (fake runtime)";
    case hipErrorInvalidDevice: return "invalid device ordinal (fake runtime)";
    case hipErrorOutOfMemory: return "out of memory (fake runtime)";
    case hipErrorHostMemoryAlreadyRegistered: return "part or all of the requested memory range is already mapped (fake runtime)";
    case hipErrorHostMemoryNotRegistered: return "pointer does not correspond to a registered memory region (fake runtime)";
    default: return "error (fake runtime)";
  }
}

hipError_t hipStreamCreateWithFlags(hipStream_t *stream, unsigned int) {
  void *s = std::malloc(8);
  std::lock_guard<std::mutex> lock(g_mutex);
  g_streams.insert(s);
  *stream = static_cast<hipStream_t>(s);
  return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!g_streams.erase(stream)) return done(hipErrorInvalidHandle);
  std::free(stream);
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }

hipError_t hipEventCreateWithFlags(hipEvent_t *event, unsigned) {
  void *e = std::malloc(8);
  std::lock_guard<std::mutex> lock(g_mutex);
  g_events.insert(e);
  *event = static_cast<hipEvent_t>(e);
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *event) { return hipEventCreateWithFlags(event, 0); }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t event) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!g_events.erase(event)) return done(hipErrorInvalidHandle);
  std::free(event);
  return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) {
  *ms = 0.0f;
  return hipSuccess;
}

hipError_t hipMalloc(void **ptr, size_t size) {
  *ptr = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (g_fail_malloc_device == t_device) return done(hipErrorOutOfMemory);
  }
  void *p = std::malloc(size ? size : 1);
  if (!p) return done(hipErrorOutOfMemory);
  std::lock_guard<std::mutex> lock(g_mutex);
  g_allocs[p] = t_device;
  *ptr = p;
  return hipSuccess;
}
hipError_t hipFree(void *ptr) {
  if (!ptr) return hipSuccess;
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!g_allocs.erase(ptr)) return done(hipErrorInvalidValue);
  std::free(ptr);
  return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t sizeBytes, hipMemcpyKind) {
  std::memcpy(dst, src, sizeBytes);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t sizeBytes, hipMemcpyKind, hipStream_t) {
  std::memcpy(dst, src, sizeBytes);
  return hipSuccess;
}
hipError_t hipMemset(void *dst, int value, size_t sizeBytes) {
  std::memset(dst, value, sizeBytes);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *dst, int value, size_t sizeBytes, hipStream_t) {
  std::memset(dst, value, sizeBytes);
  return hipSuccess;
}

hipError_t hipHostRegister(void *hostPtr, size_t sizeBytes, unsigned int) {
  const uintptr_t lo = reinterpret_cast<uintptr_t>(hostPtr), hi = lo + sizeBytes;
  if (!hostPtr || !sizeBytes) return done(hipErrorInvalidValue);
  std::lock_guard<std::mutex> lock(g_mutex);
  for (const auto &r : g_registered)
    if (lo < r.second && r.first < hi) return done(hipErrorHostMemoryAlreadyRegistered);
  g_registered[lo] = hi;
  return hipSuccess;
}
hipError_t hipHostUnregister(void *hostPtr) {
  std::lock_guard<std::mutex> lock(g_mutex);
  if (!g_registered.erase(reinterpret_cast<uintptr_t>(hostPtr))) return done(hipErrorHostMemoryNotRegistered);
  return hipSuccess;
}
hipError_t hipHostMalloc(void **ptr, size_t size, unsigned int) {
  void *p = std::malloc(size ? size : 1);
  if (!p) return done(hipErrorOutOfMemory);
  std::lock_guard<std::mutex> lock(g_mutex);
  g_registered[reinterpret_cast<uintptr_t>(p)] = reinterpret_cast<uintptr_t>(p) + (size ? size : 1);
  *ptr = p;
  return hipSuccess;
}
hipError_t hipHostFree(void *ptr) {
  if (!ptr) return hipSuccess;
  {
    std::lock_guard<std::mutex> lock(g_mutex);
    if (!g_registered.erase(reinterpret_cast<uintptr_t>(ptr))) return done(hipErrorInvalidValue);
  }
  std::free(ptr);
  return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *attributes, const void *ptr) {
  std::memset(attributes, 0, sizeof *attributes);
  const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
  std::lock_guard<std::mutex> lock(g_mutex);
  attributes->type = hipMemoryTypeUnregistered;
  for (const auto &r : g_registered)
    if (a >= r.first && a < r.second) attributes->type = hipMemoryTypeHost;
  for (const auto &r : g_allocs)
    if (ptr == r.first) attributes->type = hipMemoryTypeDevice;
  return hipSuccess;
}

}  // extern "C"
