// Development: workgroups of 256 threads resident per CU as a function of the dynamic LDS per workgroup (where are the steps?).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float *out) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[255 - threadIdx.x];
}
int main() {
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (size_t lds : {40912, 40960, 40961, 40448, 39936, 32768, 32769, 53248, 54613, 54614, 81920, 81921, 163840}) {
    int n = -1;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, lds);
    printf("lds %zu B per workgroup: %d workgroups per CU (%s)\n", lds, n, hipGetErrorString(e));
  }
  return 0;
}
