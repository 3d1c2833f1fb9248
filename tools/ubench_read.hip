// Development: what a read-only streaming kernel reaches on this chip (ceiling for values_stats /
// radix_hist).  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_read.hip -o gpurun_out/ubench_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int kInFlight, bool kNt>
__global__ __launch_bounds__(256) void read_sum(const float4 *__restrict__ p, size_t n4, float *out) {
  float acc = 0.f;
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + (kInFlight - 1) * stride < n4; i += kInFlight * stride) {
    float4 v[kInFlight];
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) {
      typedef float v4f __attribute__((ext_vector_type(4)));
      if (kNt) {
        const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p + i + u * stride));
        v[u] = make_float4(t.x, t.y, t.z, t.w);
      } else {
        v[u] = p[i + u * stride];
      }
    }
#pragma unroll
    for (int u = 0; u < kInFlight; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n4; i += stride) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) *out = acc;
}

template <int kInFlight, bool kNt>
void run(const float4 *d, size_t n4, float *out, int blocks_per_cu, int cus) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = cus * blocks_per_cu;
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((read_sum<kInFlight, kNt>), dim3(grid), dim3(256), 0, 0, d, n4, out);
  CK(hipEventRecord(a));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((read_sum<kInFlight, kNt>), dim3(grid), dim3(256), 0, 0, d, n4, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
  printf("in_flight=%d nt=%d blocks/CU=%-3d  %.3f ms  %.0f GB/s\n", kInFlight, (int)kNt, blocks_per_cu, ms, n4 * 16.0 / ms / 1e6);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const size_t n4 = 1000000000ull / 4;  // 4 GB, as 1e9 final values
  float4 *d; float *out; CK(hipMalloc(&d, n4 * 16)); CK(hipMalloc(&out, 4)); CK(hipMemset(d, 0, n4 * 16));
  const int cus = prop.multiProcessorCount;
  for (int bpc : {4, 8, 16, 32}) {
    run<1, false>(d, n4, out, bpc, cus); run<2, false>(d, n4, out, bpc, cus); run<4, false>(d, n4, out, bpc, cus);
    run<8, false>(d, n4, out, bpc, cus); run<4, true>(d, n4, out, bpc, cus);
  }
  return 0;
}
