// smmc_vector_add.hip -- the reference's GPU "hello world" (src/gpu.cu:8-47: impl_vector_add_gpu and its host
// function vector_add_gpu), which north_star names beside the Monte-Carlo engine.  Not part of the hot
// path: three n-float host arrays cross PCIe for one add per element, so the call is bound by the copies;
// the kernel itself is a 12-bytes-per-element HBM stream (16-byte accesses, grid-stride, one launch).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "smmc.h"

extern "C" __attribute__((visibility("hidden"))) int smmc_set_error_(int code, const char *message);

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void vector_add_kernel(float *__restrict__ out, const float *__restrict__ a,
                                                             const float *__restrict__ b, uint64_t n) {
  const uint64_t tid = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
  const uint64_t n4 = n / 4;  // the three device arrays start on 256-byte boundaries (one hipMalloc, n rounded up)
  const float4 *a4 = reinterpret_cast<const float4 *>(a);
  const float4 *b4 = reinterpret_cast<const float4 *>(b);
  float4 *o4 = reinterpret_cast<float4 *>(out);
  for (uint64_t i = tid; i < n4; i += stride) {
    const float4 x = a4[i], y = b4[i];
    o4[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
  const uint64_t tail = n4 * 4 + tid;
  if (tail < n) out[tail] = a[tail] + b[tail];
}

int hip_fail(const char *what, hipError_t err) {
  char buf[256];
  std::snprintf(buf, sizeof buf, "%s failed: %s", what, hipGetErrorString(err));
  return smmc_set_error_(SMMC_ERR_HIP, buf);
}

}  // namespace

extern "C" int smmc_vector_add(float *out, const float *a, const float *b, int64_t n, double *kernel_seconds) {
  if (kernel_seconds) *kernel_seconds = 0.0;
  if (n < 0 || (n > 0 && (!out || !a || !b))) return smmc_set_error_(SMMC_ERR_INVALID, "vector_add: NULL array or negative length");
  if (n == 0) return SMMC_OK;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) {
    (void)hipGetLastError();
    return smmc_set_error_(SMMC_ERR_NO_DEVICE, "no MI355X visible to this process; vector_add_gpu has no CPU fallback");
  }
  const uint64_t len = static_cast<uint64_t>(n);
  const uint64_t padded = (len + 63) / 64 * 64;  // every array on a 256-byte boundary
  float *d = nullptr;
  hipError_t err = hipMalloc(reinterpret_cast<void **>(&d), sizeof(float) * padded * 3);
  if (err != hipSuccess) return hip_fail("hipMalloc of the three device arrays", err);
  float *d_a = d, *d_b = d + padded, *d_out = d + 2 * padded;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int rc = SMMC_OK;
  do {
    if ((err = hipMemcpy(d_a, a, sizeof(float) * len, hipMemcpyHostToDevice)) != hipSuccess) { rc = hip_fail("hipMemcpy a", err); break; }
    if ((err = hipMemcpy(d_b, b, sizeof(float) * len, hipMemcpyHostToDevice)) != hipSuccess) { rc = hip_fail("hipMemcpy b", err); break; }
    if ((err = hipEventCreate(&ev0)) != hipSuccess || (err = hipEventCreate(&ev1)) != hipSuccess) { rc = hip_fail("hipEventCreate", err); break; }
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if ((err = hipGetDeviceProperties(&prop, dev)) != hipSuccess) { rc = hip_fail("hipGetDeviceProperties", err); break; }
    // enough workgroups for every CU to keep 2048 threads of 16-byte loads in flight, no more than the work
    const uint64_t want = (len / 4 + kThreads - 1) / kThreads + 1;
    const uint32_t grid = static_cast<uint32_t>(want < static_cast<uint64_t>(prop.multiProcessorCount) * 16
                                                    ? want : static_cast<uint64_t>(prop.multiProcessorCount) * 16);
    (void)hipEventRecord(ev0, nullptr);
    vector_add_kernel<<<grid, kThreads, 0, nullptr>>>(d_out, d_a, d_b, len);
    if ((err = hipGetLastError()) != hipSuccess) { rc = hip_fail("vector_add_kernel launch", err); break; }
    (void)hipEventRecord(ev1, nullptr);
    if ((err = hipEventSynchronize(ev1)) != hipSuccess) { rc = hip_fail("vector_add_kernel", err); break; }
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess && kernel_seconds) *kernel_seconds = ms * 1e-3;
    if ((err = hipMemcpy(out, d_out, sizeof(float) * len, hipMemcpyDeviceToHost)) != hipSuccess) { rc = hip_fail("hipMemcpy out", err); break; }
  } while (false);
  if (ev0) (void)hipEventDestroy(ev0);
  if (ev1) (void)hipEventDestroy(ev1);
  (void)hipFree(d);
  return rc;
}
