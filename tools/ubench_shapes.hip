// ubench_shapes.hip -- development tool: how the SHAPE of a streaming launch (not its access pattern: every variant here moves
// whole 1 KiB pieces per wave instruction, 16 bytes per lane) changes what HBM gives, for reads and for writes:
//   stride   a resident grid (256 x bpc workgroups of 256) striding over the array together
//   chunk    the same grid, each workgroup walking its own contiguous share
//   oneshot  no loop: a workgroup per 256 x U x 16 bytes, the dispatcher does the striding
// Round 4 found a plain fill at 7.0 TB/s in the one-shot shape where the resident shapes reach 5.4-6.5.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_shapes.hip -o /tmp/ubench_shapes && /tmp/ubench_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float sum4(const float4 v) { return (v.x + v.y) + (v.z + v.w); }

template <int U>
__global__ __launch_bounds__(256) void read_stride(const float4 *__restrict__ p, size_t n4, float *out) {
  float acc = 0.f;
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = i + u * stride < n4 ? p[i + u * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += sum4(v[u]);
  }
  if (acc == 123.456f) *out = acc;
}
template <int U>
__global__ __launch_bounds__(256) void read_chunk(const float4 *__restrict__ p, size_t n4, float *out) {
  float acc = 0.f;
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  for (size_t i = lo + threadIdx.x; i < hi; i += U * 256) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = i + u * 256 < hi ? p[i + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += sum4(v[u]);
  }
  if (acc == 123.456f) *out = acc;
}
template <int U>
__global__ __launch_bounds__(256) void read_oneshot(const float4 *__restrict__ p, size_t n4, float *out) {
  const size_t i = static_cast<size_t>(blockIdx.x) * (256 * U) + threadIdx.x;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = i + u * 256 < n4 ? p[i + u * 256] : make_float4(0, 0, 0, 0);
  float acc = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) acc += sum4(v[u]);
  if (acc == 123.456f) *out = acc;
}
template <int U>
__global__ __launch_bounds__(256) void write_stride(float4 *p, size_t n4) {
  const size_t stride = static_cast<size_t>(gridDim.x) * 256;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * stride < n4) p[i + u * stride] = make_float4(1.f, 2.f, 3.f, 4.f + u);
  }
}
template <int U>
__global__ __launch_bounds__(256) void write_chunk(float4 *p, size_t n4) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
  for (size_t i = lo + threadIdx.x; i < hi; i += U * 256) {
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < hi) p[i + u * 256] = make_float4(1.f, 2.f, 3.f, 4.f + u);
  }
}
template <int U>
__global__ __launch_bounds__(256) void write_oneshot(float4 *p, size_t n4) {
  const size_t i = static_cast<size_t>(blockIdx.x) * (256 * U) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < n4) p[i + u * 256] = make_float4(1.f, 2.f, 3.f, 4.f + u);
}

template <typename F>
void timed(const char *what, int u, int bpc, size_t n4, F launch) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int w = 0; w < 2; ++w) launch();
  float best = 1e9f, sum = 0.f;
  const int reps = 6;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); sum += ms; if (ms < best) best = ms;
  }
  printf("%-14s U=%d bpc=%-3d bytes=%.3g  mean %.4f ms %.0f GB/s   best %.4f ms %.0f GB/s\n", what, u, bpc, n4 * 16.0, sum / reps,
         n4 * 16.0 / (sum / reps) / 1e6, best, n4 * 16.0 / best / 1e6);
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

template <int U>
void sweep(float4 *d, float *out, size_t n4) {
  for (int bpc : {4, 8, 16}) {
    const int grid = 256 * bpc;
    timed("read stride", U, bpc, n4, [&] { hipLaunchKernelGGL(read_stride<U>, dim3(grid), dim3(256), 0, 0, d, n4, out); });
    timed("read chunk", U, bpc, n4, [&] { hipLaunchKernelGGL(read_chunk<U>, dim3(grid), dim3(256), 0, 0, d, n4, out); });
    timed("write stride", U, bpc, n4, [&] { hipLaunchKernelGGL(write_stride<U>, dim3(grid), dim3(256), 0, 0, d, n4); });
    timed("write chunk", U, bpc, n4, [&] { hipLaunchKernelGGL(write_chunk<U>, dim3(grid), dim3(256), 0, 0, d, n4); });
  }
  const unsigned grid1 = static_cast<unsigned>((n4 + 256 * U - 1) / (256 * U));
  timed("read oneshot", U, 0, n4, [&] { hipLaunchKernelGGL(read_oneshot<U>, dim3(grid1), dim3(256), 0, 0, d, n4, out); });
  timed("write oneshot", U, 0, n4, [&] { hipLaunchKernelGGL(write_oneshot<U>, dim3(grid1), dim3(256), 0, 0, d, n4); });
}

int main() {
  const size_t n4_big = 361000000ull, n4_small = 100000000ull / 4;  // 5.776 GB (keepdata's 4e6 x 361), 400 MB (1e8 final values)
  float4 *d; float *out; CK(hipMalloc(&d, n4_big * 16)); CK(hipMalloc(&out, 4)); CK(hipMemset(d, 0, n4_big * 16));
  for (size_t n4 : {n4_big, n4_small}) {
    sweep<1>(d, out, n4); sweep<2>(d, out, n4); sweep<4>(d, out, n4); sweep<8>(d, out, n4);
  }
  return 0;
}
