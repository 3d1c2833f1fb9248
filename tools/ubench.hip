// ubench.hip -- instruction issue-rate probes for gfx950 (development tool, not product).
// Measures wave64 VALU throughput per CU for the instructions the paths kernel leans on.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench.hip -o gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 4096;
constexpr int kUnroll = 16;
#define REP4(S) S "\n" S "\n" S "\n" S
static_assert(kUnroll == 16, "REP4 of a four-instruction pattern");

// shader clock: (memtime ticks, 100 MHz realtime ticks) per workgroup
template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned *out, unsigned seed, unsigned long long *clk) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 12345u, d = b + 777u;
  unsigned long long w0 = a, w1 = b, w2 = c, w3 = d;
  float f0 = a * 1e-9f + 1.0f, f1 = 1.0001f, f2 = 0.5f, f3 = 1.5f;
  for (int i = 0; i < kIters; ++i) {
    {
      // ONE asm statement per 16 instructions: after every asm statement hipcc's hazard recogniser adds an s_nop
      // (round 1's version of this tool had one per four instructions)
      if constexpr (OP == 0) {  // v_xor_b32 (4 independent chains)
        asm volatile(REP4("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if constexpr (OP == 1) {  // v_mul_lo_u32
        asm volatile(REP4("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if constexpr (OP == 2) {  // v_mul_hi_u32
        asm volatile(REP4("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if constexpr (OP == 3) {  // v_mad_u64_u32
        asm volatile(REP4("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %6, %5, 0\n"
                     " v_mad_u64_u32 %2, vcc, %7, %5, 0\n v_mad_u64_u32 %3, vcc, %8, %5, 0")
                     : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(a), "v"(seed), "v"(b), "v"(c), "v"(d) : "vcc");
        a ^= (unsigned)w0; b ^= (unsigned)w1;  // keep live (2 extra xor per 16 mads)
      } else if constexpr (OP == 4) {  // v_fma_f32
        asm volatile(REP4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5")
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(1.0000001f), "v"(1e-9f));
      } else if constexpr (OP == 5) {  // v_mul_u32_u24
        asm volatile(REP4("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if constexpr (OP == 6) {  // v_mul_hi_u32_u24
        asm volatile(REP4("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if constexpr (OP == 7) {  // v_sqrt_f32
        asm volatile(REP4("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3")
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
      } else if constexpr (OP == 8) {  // v_cvt_f32_u32
        asm volatile(REP4("v_cvt_f32_u32 %0, %4\n v_cvt_f32_u32 %1, %5\n v_cvt_f32_u32 %2, %6\n v_cvt_f32_u32 %3, %7")
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b), "v"(c), "v"(d));
      } else if constexpr (OP == 9) {  // v_pk_fma_f32 (2 fma per lane per instruction)
        asm volatile(REP4("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3")
                     : "+v"(w0), "+v"(w1) : "v"(w2), "v"(w3));
      } else if constexpr (OP == 10) {  // v_mul_f32 + v_fmac x2 dependent chain (the div100 step)
        asm volatile(REP4("v_mul_f32 %1, 0x3c23d70a, %0\n v_fmac_f32 %0, 0xc2c80000, %1\n v_fmac_f32 %1, 0x3c23d70a, %0\n v_mul_f32 %0, %1, %2")
                     : "+v"(f0), "+v"(f1) : "v"(f2));
      } else if constexpr (OP == 11) {  // v_mad_u64_u32 alone: four independent products of loop-invariant inputs (pure issue rate)
        asm volatile(REP4("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %6, %5, 0\n"
                     " v_mad_u64_u32 %2, vcc, %7, %5, 0\n v_mad_u64_u32 %3, vcc, %8, %5, 0")
                     : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "v"(a), "v"(seed), "v"(b), "v"(c), "v"(d) : "vcc");
      } else if constexpr (OP == 12) {  // v_bitop3_b32 (three-input XOR, one scalar operand: Philox's round XORs)
        asm volatile(REP4("v_bitop3_b32 %0, %0, %1, %4 bitop3:0x96\n v_bitop3_b32 %1, %1, %2, %4 bitop3:0x96\n"
                     " v_bitop3_b32 %2, %2, %3, %4 bitop3:0x96\n v_bitop3_b32 %3, %3, %0, %4 bitop3:0x96")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));
      } else if constexpr (OP == 13) {  // v_and_b32 SDWA reading the upper half-word (the radius bin / sector offset of stream v3)
        asm volatile(REP4("v_and_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
                     " v_and_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
                     " v_and_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
                     " v_and_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));
      } else if constexpr (OP == 14) {  // v_cvt_f32_i32 (the signed distance of stream v3's radius)
        asm volatile(REP4("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %6\n v_cvt_f32_i32 %3, %7")
                     : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(a), "v"(b), "v"(c), "v"(d));
      } else if constexpr (OP == 15) {  // v_and_or_b32 (low bits into the mantissa of 1.0f: the residual angle)
        asm volatile(REP4("v_and_or_b32 %0, %0, %4, 1.0\n v_and_or_b32 %1, %1, %4, 1.0\n v_and_or_b32 %2, %2, %4, 1.0\n v_and_or_b32 %3, %3, %4, 1.0")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));
      } else if constexpr (OP == 16) {  // v_lshl_add_u32 (LDS address of a table draw)
        asm volatile(REP4("v_lshl_add_u32 %0, %0, 2, %4\n v_lshl_add_u32 %1, %1, 2, %4\n v_lshl_add_u32 %2, %2, 2, %4\n v_lshl_add_u32 %3, %3, 2, %4")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed));
      } else if constexpr (OP == 17) {  // v_mul_f32 with a 32-bit literal (the divide shortcut's low product)
        asm volatile(REP4("v_mul_f32 %0, 0x2f75c28f, %0\n v_mul_f32 %1, 0x2f75c28f, %1\n v_mul_f32 %2, 0x2f75c28f, %2\n v_mul_f32 %3, 0x2f75c28f, %3")
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (unsigned)w0 ^ (unsigned)w1 ^ (unsigned)w2 ^ (unsigned)w3 ^
                                               __float_as_uint(f0 + f1 + f2 + f3);
}

template <int OP>
int run(const char *name, int waves_per_simd, unsigned *d_out, int cus, unsigned long long *d_clk) {
  // blocks of 256 threads = 4 waves = 1 wave per SIMD; `waves_per_simd` blocks per CU
  int grid = cus * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, d_out, 1u, d_clk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, d_out, 2u, d_clk);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * grid);
  CK(hipMemcpy(h.data(), d_clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
  double ticks = 0, real = 0;
  for (int i = 0; i < grid; ++i) { ticks += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
  const double ghz = ticks / real * 0.1;  // memrealtime = 100 MHz
  const double clk_per_inst = (ticks / grid) / ((double)kIters * kUnroll) / waves_per_simd;  // per SIMD: waves share it
  double insts_per_wave = (double)kIters * kUnroll;  // target instructions per wave
  double waves = (double)grid * 4;
  double total = insts_per_wave * waves;
  double per_simd_per_s = total / (ms * 1e-3) / (cus * 4.0);
  printf("%-22s waves/SIMD=%d  %8.3f ms  %7.2f Ginst/s/SIMD  in-kernel clock %.2f GHz  => %5.2f shader-clk/inst/SIMD\n", name,
         waves_per_simd, ms, per_simd_per_s * 1e-9, ghz, clk_per_inst);
  // machine-readable twin (tools/valu_weights.py): clk_per_inst = 1 / (Ginst/s/SIMD) * in-kernel GHz
  printf("{\"probe\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ginst_per_s_per_simd\": %.4f, \"clock_ghz\": %.3f, "
         "\"clk_per_inst\": %.4f}\n", name, waves_per_simd, ms, per_simd_per_s * 1e-9, ghz, ghz / (per_simd_per_s * 1e-9));
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  int cus = p.multiProcessorCount;
  printf("%s CUs=%d clock=%d kHz\n", p.gcnArchName, cus, p.clockRate);
  unsigned *d_out;
  CK(hipMalloc(&d_out, sizeof(unsigned) * cus * 8 * 256));
  unsigned long long *d_clk;
  CK(hipMalloc(&d_clk, sizeof(unsigned long long) * 2 * cus * 8));
  for (int w : {2, 4, 8}) {
    run<0>("v_xor_b32", w, d_out, cus, d_clk);
    run<1>("v_mul_lo_u32", w, d_out, cus, d_clk);
    run<2>("v_mul_hi_u32", w, d_out, cus, d_clk);
    run<3>("v_mad_u64_u32(+.125xor)", w, d_out, cus, d_clk);
    run<11>("v_mad_u64_u32", w, d_out, cus, d_clk);
    run<12>("v_bitop3_b32", w, d_out, cus, d_clk);
    run<4>("v_fma_f32", w, d_out, cus, d_clk);
    run<17>("v_mul_f32(literal)", w, d_out, cus, d_clk);
    run<13>("v_and_b32_sdwa", w, d_out, cus, d_clk);
    run<14>("v_cvt_f32_i32", w, d_out, cus, d_clk);
    run<15>("v_and_or_b32", w, d_out, cus, d_clk);
    run<16>("v_lshl_add_u32", w, d_out, cus, d_clk);
    run<5>("v_mul_u32_u24", w, d_out, cus, d_clk);
    run<6>("v_mul_hi_u32_u24", w, d_out, cus, d_clk);
    run<7>("v_sqrt_f32", w, d_out, cus, d_clk);
    run<8>("v_cvt_f32_u32", w, d_out, cus, d_clk);
    run<9>("v_pk_fma_f32", w, d_out, cus, d_clk);
    run<10>("div100 chain(4 inst)", w, d_out, cus, d_clk);
  }
  return 0;
}
