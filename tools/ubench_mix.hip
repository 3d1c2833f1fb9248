// ubench_mix.hip -- the issue ceiling of paths_kernel's OWN instruction mix (development tool, not product).
//
// One Philox block of paths_kernel<gaussian> is 70 VALU instructions (tools/isa_loop_count.py): 16
// v_mad_u64_u32, 15 v_bitop3_b32 + 3 v_xor_b32, 28 binary32 (10 v_fma, 8 v_fmac, 8 v_mul, 2 v_fmamk), 2
// v_cvt_f32_i32, 4 v_and_b32 (SDWA), 2 v_and_or_b32; the table kernel's block (8 periods) is 84: 26
// v_mad_u64_u32, 4 v_mul_hi_u32, 15 v_bitop3_b32 + 3 v_xor_b32, 16 v_mul_f32 + 8 v_fmac_f32, 8 v_lshl_add_u32,
// 4 v_mov_b32.  These probes issue exactly those mixes, in the kernel's order of kinds, with NO memory
// access and only short dependences (four register sets in rotation), at 2 / 4 / 6 / 8 waves per SIMD: the
// time per block is what the VALU needs for the mix when nothing else stands in the way.  The kernel's
// measured time per block divided by this is how close it runs to its instruction mix's own ceiling.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mix.hip -o <exe>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kBlocks = 2048;  // mix blocks per wave

struct Regs {
  unsigned long long w[4];
  unsigned a[4];
  float f[4];
};

// One asm statement per block (named operands): between separate asm statements hipcc's hazard recogniser
// inserts an s_nop it does not need between ordinary VALU instructions (the kernel's own loop has none).
#define MAD(i, j) "v_mad_u64_u32 %[w" #i "], vcc, %[a" #j "], %[k], 0\n"
#define BITOP(i, j) "v_bitop3_b32 %[a" #i "], %[a" #i "], %[a" #j "], %[k] bitop3:0x96\n"
#define XOR(i) "v_xor_b32 %[a" #i "], %[k], %[a" #i "]\n"
#define FMA(i, j) "v_fma_f32 %[f" #i "], %[f" #i "], %[f" #j "], %[kf]\n"
#define FMAC(i, j) "v_fmac_f32 %[f" #i "], 0x3c23d70a, %[f" #j "]\n"
#define MUL(i, j) "v_mul_f32 %[f" #i "], 0x2f75c28f, %[f" #j "]\n"
#define MULV(i, j) "v_mul_f32 %[f" #i "], %[f" #i "], %[f" #j "]\n"
#define FMAMK(i, j) "v_fmamk_f32 %[f" #i "], %[f" #j "], 0x3d490fdb, %[f" #i "]\n"
#define CVT(i, j) "v_cvt_f32_i32 %[f" #i "], %[a" #j "]\n"
#define SDWA(i, j) "v_and_b32_sdwa %[a" #i "], %[a" #j "], %[k] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define ANDOR(i, j) "v_and_or_b32 %[a" #i "], %[a" #j "], %[k], 1.0\n"
#define MULHI(i, j) "v_mul_hi_u32 %[a" #i "], %[a" #j "], %[k]\n"
#define LSHLADD(i, j) "v_lshl_add_u32 %[a" #i "], %[a" #j "], 2, %[k]\n"
#define MOV(i, j) "v_mov_b32 %[a" #i "], %[a" #j "]\n"
#define OPERANDS                                                                                                     \
  : [w0] "+v"(r.w[0]), [w1] "+v"(r.w[1]), [w2] "+v"(r.w[2]), [w3] "+v"(r.w[3]), [a0] "+v"(r.a[0]), [a1] "+v"(r.a[1]), \
    [a2] "+v"(r.a[2]), [a3] "+v"(r.a[3]), [f0] "+v"(r.f[0]), [f1] "+v"(r.f[1]), [f2] "+v"(r.f[2]), [f3] "+v"(r.f[3]) \
  : [k] "s"(k), [kf] "s"(kf)                                                                                          \
  : "vcc"
// rounds 0-1 (what is left of them per lane), then eight rounds of 2 mads + 2 three-input XORs: 16 mads, 15 + 3 XORs
#define PHILOX                                                                                                         \
  XOR(0) MAD(0, 0) XOR(1) XOR(2) MAD(1, 1) MAD(2, 2)                                                                   \
  BITOP(0, 1) BITOP(2, 3) MAD(0, 0) MAD(1, 2) BITOP(0, 1) BITOP(2, 3) MAD(1, 0) MAD(2, 2) BITOP(0, 1) BITOP(2, 3)      \
  MAD(2, 0) MAD(3, 2) BITOP(0, 1) BITOP(2, 3) MAD(3, 0) MAD(0, 2) BITOP(0, 1) BITOP(2, 3) MAD(0, 0) MAD(1, 2)          \
  BITOP(0, 1) BITOP(2, 3) MAD(1, 0) MAD(2, 2) BITOP(0, 1) BITOP(2, 3) MAD(0, 0) BITOP(1, 2)

// paths_kernel<gaussian, fast divide>: Philox, the draw (masks, conversions, cubics, rotations), four steps
__device__ __forceinline__ void gaussian_block(Regs &r, unsigned k, float kf) {
  asm volatile(PHILOX
               SDWA(0, 1) ANDOR(1, 2) CVT(0, 3) CVT(1, 0) SDWA(2, 3) ANDOR(3, 0) SDWA(0, 2) SDWA(1, 3) FMAMK(2, 0) FMAMK(3, 1)
               FMA(0, 1) FMAC(1, 2) FMA(2, 3) FMA(3, 0) FMAC(0, 1) FMA(1, 2) FMA(2, 3) FMA(3, 0) FMAC(0, 1)
               FMA(1, 2) MULV(0, 1) FMAC(2, 3) MUL(3, 0) FMA(1, 2) FMAC(3, 0) MULV(0, 3) MUL(1, 0) FMA(2, 3)
               FMAC(1, 0) MULV(0, 1) MUL(3, 0) FMA(2, 1) FMAC(3, 0) MULV(2, 3) MUL(1, 2) FMAC(1, 2)
               OPERANDS);
}
// paths_kernel<table dense, fast divide>: the same Philox, eight base-T digits, eight gather addresses, eight steps
#define DIGITS MAD(0, 1) MAD(1, 2) MAD(2, 3) MAD(3, 0) MAD(0, 2) MULHI(1, 3) MULHI(2, 0) MOV(3, 1) MOV(0, 2)
#define STEP(i, j, l) MULV(i, j) MUL(l, i) FMAC(i, l)
__device__ __forceinline__ void table_block(Regs &r, unsigned k, float kf) {
  asm volatile(PHILOX DIGITS DIGITS
               LSHLADD(0, 1) LSHLADD(1, 2) LSHLADD(2, 3) LSHLADD(3, 0) LSHLADD(0, 1) LSHLADD(1, 2) LSHLADD(2, 3) LSHLADD(3, 0)
               STEP(0, 1, 2) STEP(1, 2, 3) STEP(2, 3, 0) STEP(3, 0, 1) STEP(0, 1, 2) STEP(1, 2, 3) STEP(2, 3, 0) STEP(3, 0, 1)
               OPERANDS);
}

template <int MIX>
__global__ __launch_bounds__(256) void probe(unsigned *out, unsigned k, float kf, unsigned long long *clk, int n_blocks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  Regs r;
  for (int i = 0; i < 4; ++i) {
    r.a[i] = threadIdx.x * 2654435761u + k + i;
    r.w[i] = r.a[i];
    r.f[i] = 1.0f + i * 0.25f;
  }
  for (int i = 0; i < n_blocks; ++i) {
    if constexpr (MIX == 0) gaussian_block(r, k, kf);
    else table_block(r, k, kf);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r.a[0] ^ r.a[1] ^ r.a[2] ^ r.a[3] ^ (unsigned)r.w[0] ^ (unsigned)r.w[1] ^ (unsigned)r.w[2] ^
                                               (unsigned)r.w[3] ^ __float_as_uint(r.f[0] + r.f[1] + r.f[2] + r.f[3]);
}

template <int MIX>
int run(const char *name, int insts, int waves_per_simd, unsigned *d_out, int cus, unsigned long long *d_clk) {
  const int grid = cus * waves_per_simd;  // 256 threads = one wave per SIMD
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(probe<MIX>, dim3(grid), dim3(256), 0, 0, d_out, 3u, 1.5f, d_clk, kBlocks);
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(probe<MIX>, dim3(grid), dim3(256), 0, 0, d_out, 5u, 1.5f, d_clk, kBlocks);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<unsigned long long> h(2 * grid);
  CK(hipMemcpy(h.data(), d_clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
  double ticks = 0, real = 0;
  for (int i = 0; i < grid; ++i) { ticks += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
  const double ghz = ticks / real * 0.1;
  // blocks per second per SIMD, and the clocks one block occupies a SIMD's VALU for
  const double blocks_per_s_simd = (double)kBlocks * waves_per_simd / (ms * 1e-3);
  const double clk_per_block = ghz * 1e9 / blocks_per_s_simd;
  printf("{\"probe\": \"%s\", \"valu_insts_per_block\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"clock_ghz\": %.3f, "
         "\"clk_per_block\": %.2f, \"ns_per_block_per_simd\": %.3f, \"slots_per_inst\": %.4f}\n",
         name, insts, waves_per_simd, ms, ghz, clk_per_block, 1e9 / blocks_per_s_simd, clk_per_block / 2.0 / insts);
  return 0;
}

// Sustained load: launches as long as paths_kernel's (1e8 x 360 Gaussian paths are 137 329 blocks per SIMD: 34 332 per
// wave at four waves per SIMD, ~13 ms), back to back for a quarter of a second: which clock does the chip HOLD under
// this instruction mix, and how long does a block take then?
template <int MIX>
int sustained(const char *name, int insts, int waves_per_simd, int n_blocks, int launches, unsigned *d_out, int cus, unsigned long long *d_clk) {
  const int grid = cus * waves_per_simd;
  std::vector<unsigned long long> h(2 * grid);
  for (int l = 0; l < launches; ++l) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MIX>, dim3(grid), dim3(256), 0, 0, d_out, 5u, 1.5f, d_clk, n_blocks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d_clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    double ticks = 0, real = 0;
    for (int i = 0; i < grid; ++i) { ticks += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    const double ghz = ticks / real * 0.1;
    const double ns_per_block = ms * 1e6 / ((double)n_blocks * waves_per_simd);
    if (l < 3 || l % 8 == 7 || l + 1 == launches)
      printf("{\"probe\": \"%s sustained\", \"valu_insts_per_block\": %d, \"waves_per_simd\": %d, \"launch\": %d, \"ms\": %.3f, "
             "\"clock_ghz\": %.3f, \"ns_per_block_per_simd\": %.3f, \"clk_per_block\": %.2f}\n", name, insts, waves_per_simd, l, ms, ghz,
             ns_per_block, ns_per_block * ghz);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
  }
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, cus, p.clockRate);
  unsigned *d_out;
  CK(hipMalloc(&d_out, sizeof(unsigned) * cus * 8 * 256));
  unsigned long long *d_clk;
  CK(hipMalloc(&d_clk, sizeof(unsigned long long) * 2 * cus * 8));
  for (int w : {2, 4, 6, 8}) {
    if (run<0>("mix_gaussian_block", 70, w, d_out, cus, d_clk)) return 1;
    if (run<1>("mix_table_block", 84, w, d_out, cus, d_clk)) return 1;
  }
  if (sustained<0>("mix_gaussian_block", 70, 4, 34332, 24, d_out, cus, d_clk)) return 1;
  if (sustained<1>("mix_table_block", 84, 8, 8583, 24, d_out, cus, d_clk)) return 1;
  return 0;
}
