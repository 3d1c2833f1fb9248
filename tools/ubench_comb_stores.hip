// Micro-benchmark behind the comb keepdata kernel's store phase: the kernel's
// exact store pattern (64 streams per wave, whole 128-byte lines, 8 streams per wave-wide 16-byte
// store, 8 such stores per tile) with NOTHING else -- as a function of the waves resident per CU, of a
// gap of dependent VALU work between two tiles (the compute a real wave does there) and of how many
// tiles' stores a wave issues back to back.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_comb_stores.hip -o ubench_comb_stores && ./ubench_comb_stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// kStatic: chunk c of wave w is w + k * (all waves) instead of the next number from a counter
template <int kTilesPerBurst, bool kStatic = false>
__global__ __launch_bounds__(1024) void comb_stores(float *out, unsigned long long n_rows, unsigned row_len, int gap,
                                                    unsigned long long *next) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned sub = lane / 8, quad = lane % 8;
  const unsigned long long base_f = reinterpret_cast<uintptr_t>(out) >> 2;
  const unsigned long long n_chunks = (n_rows / 2048) * 32;  // one row per stream
  float acc = 1.0f + lane;
  const unsigned waves_per_block = blockDim.x >> 6;
  unsigned long long c_static = (unsigned long long)blockIdx.x * waves_per_block + (threadIdx.x >> 6);
  for (;;) {
    unsigned long long c = 0;
    if (kStatic) {
      c = c_static;
      c_static += (unsigned long long)gridDim.x * waves_per_block;
    } else {
      if (lane == 0) c = atomicAdd(next, 1ull);
    }
    c = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(c >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)c);
    if (c >= n_chunks) break;
    const unsigned long long row0 = (c / 32) * 2048 + (c % 32);
    const unsigned phi = (unsigned)(base_f + row0 * row_len) & 31u;
    const unsigned first_t = (phi != 0 && row0 != 0) ? 1u : 0u;
    const unsigned n_t = (phi + row_len + 31) / 32;
    for (unsigned t0 = first_t; t0 < n_t; t0 += kTilesPerBurst) {
      for (int g = 0; g < gap * kTilesPerBurst; ++g) acc = __builtin_fmaf(acc, 1.0000001f, 0.5f);  // dependent chain
#pragma unroll
      for (unsigned tt = 0; tt < kTilesPerBurst; ++tt) {
        const unsigned t = t0 + tt;
        if (t < n_t) {
#pragma unroll
          for (unsigned it = 0; it < 8; ++it) {
            const unsigned l = sub + 8 * it;
            const long long a = (long long)((row0 + 32ull * l) * row_len) - phi + 32ll * t + 4 * quad;
            if (a >= 0 && (unsigned long long)a + 4 <= n_rows * row_len)
              *reinterpret_cast<float4 *>(out + a) = make_float4(acc, 2.0f, 3.0f, 4.0f + l);
          }
        }
      }
    }
  }
}

int main() {
  const unsigned long long n_rows = 4000000;
  const unsigned row_len = 361;
  float *d;
  unsigned long long *next;
  CK(hipMalloc(&d, n_rows * row_len * 4ull + 1024));
  CK(hipMalloc(&next, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = 4.0 * (n_rows / 2048 * 2048) * row_len;
  // shape: one workgroup of 64 W threads per CU (what the product launches) or W / 4 workgroups of 256
  for (int small_blocks : {0, 1}) {
    for (int is_static : {0, 1}) {
      for (int waves : {8, 12, 16, 32}) {
        if (!small_blocks && waves > 16) continue;
        for (int gap : {0, 200}) {
          float best = 1e9f;
          const dim3 grid(small_blocks ? 256 * waves / 4 : 256), block(small_blocks ? 256 : 64 * waves);
          for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemsetAsync(next, 0, 8, 0));
            CK(hipEventRecord(e0));
            if (is_static) hipLaunchKernelGGL((comb_stores<1, true>), grid, block, 0, 0, d, n_rows, row_len, gap, next);
            else hipLaunchKernelGGL((comb_stores<1, false>), grid, block, 0, 0, d, n_rows, row_len, gap, next);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
          }
          printf("%s, %s chunks, %2d waves per CU, gap %3d fma per tile: %.3f ms  %.0f GB/s\n",
                 small_blocks ? "workgroups of 256" : "one workgroup per CU", is_static ? "static " : "counter", waves, gap, best,
                 bytes / best / 1e6);
        }
      }
    }
  }
  for (int burst : {1, 2, 4}) {
    for (int waves : {8, 12, 13, 16}) {
      for (int gap : {0, 100, 200, 400, 800}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
          CK(hipMemsetAsync(next, 0, 8, 0));
          CK(hipEventRecord(e0));
          if (burst == 1) hipLaunchKernelGGL(comb_stores<1>, dim3(256), dim3(64 * waves), 0, 0, d, n_rows, row_len, gap, next);
          if (burst == 2) hipLaunchKernelGGL(comb_stores<2>, dim3(256), dim3(64 * waves), 0, 0, d, n_rows, row_len, gap, next);
          if (burst == 4) hipLaunchKernelGGL(comb_stores<4>, dim3(256), dim3(64 * waves), 0, 0, d, n_rows, row_len, gap, next);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep && ms < best) best = ms;
        }
        printf("tiles per burst %d, %2d waves per CU, gap %3d fma per tile: %.3f ms  %.0f GB/s\n", burst, waves, gap, best, bytes / best / 1e6);
      }
    }
  }
  return 0;
}
