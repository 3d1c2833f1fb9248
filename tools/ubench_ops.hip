// ubench_ops.hip -- issue rate of single VALU opcodes on gfx950 BY OPERAND KIND (development tool, not product).
//
// tools/ubench.hip (round 1, refreshed in round 4) found v_bitop3_b32, v_and_or_b32, v_lshl_add_u32 and the SDWA
// v_and_b32 at HALF the rate of v_xor_b32 / v_fma_f32 in probes that read one SGPR operand.  This tool separates
// the two possible causes -- the opcode, or the scalar operand -- by timing each opcode with VGPR-only operands,
// with an SGPR operand, and with an inline constant / literal, at 4 and 8 waves per SIMD.  Every probe is a loop
// of 64-instruction asm blocks: 4 independent register chains x 16, no memory access.
// Output: one JSON line per probe.  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_ops.hip -o <exe>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 1024;
constexpr int kPerIter = 64;

#define REP4(S) S S S S
#define REP16(S) REP4(REP4(S))
// one instruction template applied to the four register sets; $ is replaced by the set's digit through the macros below
#define OPERANDS                                                                                                     \
  : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [w3] "+v"(w3), [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), \
    [b0] "+v"(b0), [b1] "+v"(b1), [b2] "+v"(b2), [b3] "+v"(b3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3)  \
  : [sk] "s"(sk), [sf] "s"(sf), [vk] "v"(vk), [vf] "v"(vf), [vg] "v"(vg)                                                     \
  : "vcc", "s10", "s11", "s12", "s13", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51"

// I(n): the instruction for register set n (a string); the probe body is I(0) I(1) I(2) I(3) sixteen times
#define PROBE(ID, I)                                              \
  else if constexpr (OP == ID) {                                  \
    asm volatile(REP16(I(0) I(1) I(2) I(3)) OPERANDS);            \
  }

#define XOR_V(n) "v_xor_b32 %[a" #n "], %[vk], %[a" #n "]\n"
#define XOR_S(n) "v_xor_b32 %[a" #n "], %[sk], %[a" #n "]\n"
#define XOR_L(n) "v_xor_b32 %[a" #n "], 0x9e3779b9, %[a" #n "]\n"
#define XOR_E64_S(n) "v_xor_b32_e64 %[a" #n "], %[a" #n "], %[sk]\n"
#define ADD_V(n) "v_add_u32 %[a" #n "], %[vk], %[a" #n "]\n"
#define ADD_S(n) "v_add_u32 %[a" #n "], %[sk], %[a" #n "]\n"
#define LSHR_I(n) "v_lshrrev_b32 %[a" #n "], 7, %[a" #n "]\n"
#define BITOP_V(n) "v_bitop3_b32 %[a" #n "], %[a" #n "], %[b" #n "], %[vk] bitop3:0x96\n"
#define BITOP_S(n) "v_bitop3_b32 %[a" #n "], %[a" #n "], %[b" #n "], %[sk] bitop3:0x96\n"
#define ANDOR_V(n) "v_and_or_b32 %[a" #n "], %[a" #n "], %[vk], %[b" #n "]\n"
#define ANDOR_S(n) "v_and_or_b32 %[a" #n "], %[a" #n "], %[sk], 1.0\n"
#define LSHLADD_V(n) "v_lshl_add_u32 %[a" #n "], %[a" #n "], 2, %[vk]\n"
#define LSHLADD_S(n) "v_lshl_add_u32 %[a" #n "], %[a" #n "], 2, %[sk]\n"
#define ADD3_V(n) "v_add3_u32 %[a" #n "], %[a" #n "], %[b" #n "], %[vk]\n"
#define OR3_V(n) "v_or3_b32 %[a" #n "], %[a" #n "], %[b" #n "], %[vk]\n"
#define XAD_V(n) "v_xad_u32 %[a" #n "], %[a" #n "], %[b" #n "], %[vk]\n"
#define BFE_I(n) "v_bfe_u32 %[a" #n "], %[a" #n "], 3, 19\n"
#define BFI_V(n) "v_bfi_b32 %[a" #n "], %[vk], %[a" #n "], %[b" #n "]\n"
#define ALIGNBIT_V(n) "v_alignbit_b32 %[a" #n "], %[a" #n "], %[b" #n "], 13\n"
#define PERM_V(n) "v_perm_b32 %[a" #n "], %[a" #n "], %[b" #n "], %[vk]\n"
#define SDWA_V(n) "v_and_b32_sdwa %[a" #n "], %[a" #n "], %[vk] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define SDWA_S(n) "v_and_b32_sdwa %[a" #n "], %[a" #n "], %[sk] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define MAD64_V(n) "v_mad_u64_u32 %[w" #n "], vcc, %[a" #n "], %[vk], 0\n"
#define MAD64_S(n) "v_mad_u64_u32 %[w" #n "], vcc, %[a" #n "], %[sk], 0\n"
#define MAD64_SGPRCARRY(n) "v_mad_u64_u32 %[w" #n "], s[10:11], %[a" #n "], %[sk], 0\n"
#define MAD64_ADD(n) "v_mad_u64_u32 %[w" #n "], vcc, %[a" #n "], %[vk], %[w" #n "]\n"
#define MULLO_V(n) "v_mul_lo_u32 %[a" #n "], %[a" #n "], %[vk]\n"
#define MULHI_V(n) "v_mul_hi_u32 %[a" #n "], %[a" #n "], %[vk]\n"
#define MULHI_S(n) "v_mul_hi_u32 %[a" #n "], %[a" #n "], %[sk]\n"
#define MAD24_V(n) "v_mad_u32_u24 %[a" #n "], %[a" #n "], %[vk], %[b" #n "]\n"
#define FMA_V(n) "v_fma_f32 %[f" #n "], %[f" #n "], %[vf], %[vf]\n"
#define FMA_S(n) "v_fma_f32 %[f" #n "], %[f" #n "], %[vf], %[sf]\n"
#define FMAC_V(n) "v_fmac_f32 %[f" #n "], %[vf], %[f" #n "]\n"
#define FMAC_L(n) "v_fmac_f32 %[f" #n "], 0x3c23d70a, %[f" #n "]\n"
#define FMAMK(n) "v_fmamk_f32 %[f" #n "], %[f" #n "], 0x3d490fdb, %[vf]\n"
#define MUL_V(n) "v_mul_f32 %[f" #n "], %[vf], %[f" #n "]\n"
#define MUL_S(n) "v_mul_f32 %[f" #n "], %[sf], %[f" #n "]\n"
#define MUL_L(n) "v_mul_f32 %[f" #n "], 0x2f75c28f, %[f" #n "]\n"
#define ADDF_V(n) "v_add_f32 %[f" #n "], %[vf], %[f" #n "]\n"
#define CVT_I32(n) "v_cvt_f32_i32 %[f" #n "], %[a" #n "]\n"
#define CVT_U32(n) "v_cvt_f32_u32 %[f" #n "], %[a" #n "]\n"
#define MOV_V(n) "v_mov_b32 %[a" #n "], %[b" #n "]\n"
#define CNDMASK(n) "v_cndmask_b32 %[a" #n "], %[a" #n "], %[b" #n "], vcc\n"
#define CMP(n) "v_cmp_lt_u32 vcc, %[a" #n "], %[vk]\n"
#define PKFMA(n) "v_pk_fma_f32 %[w" #n "], %[w" #n "], %[w" #n "], %[w" #n "]\n"
#define PKMUL(n) "v_pk_mul_f32 %[w" #n "], %[w" #n "], %[w" #n "]\n"
#define FMA64(n) "v_fma_f64 %[w" #n "], %[w" #n "], %[w" #n "], %[w" #n "]\n"
#define LSHL64(n) "v_lshlrev_b64 %[w" #n "], 3, %[w" #n "]\n"
// explicit registers (clobbered): do the banks (register index mod 4) of the sources matter?
// (registers v40 .. v51 only, all named as clobbered; the macro's argument picks one of four fixed triples)
#define X3_0(op, tail) op " v40, v40, v45, v46" tail "\n"   /* banks 0, 1, 2 */
#define X3_1(op, tail) op " v41, v41, v46, v47" tail "\n"   /* 1, 2, 3 */
#define X3_2(op, tail) op " v42, v42, v47, v44" tail "\n"   /* 2, 3, 0 */
#define X3_3(op, tail) op " v43, v43, v44, v45" tail "\n"   /* 3, 0, 1 */
#define S3_0(op, tail) op " v40, v40, v44, v48" tail "\n"   /* one bank each */
#define S3_1(op, tail) op " v41, v41, v45, v49" tail "\n"
#define S3_2(op, tail) op " v42, v42, v46, v50" tail "\n"
#define S3_3(op, tail) op " v43, v43, v47, v51" tail "\n"
#define T2_0(op, tail) op " v40, v40, v44, v45" tail "\n"   /* two of the three sources in one bank */
#define T2_1(op, tail) op " v41, v41, v45, v46" tail "\n"
#define T2_2(op, tail) op " v42, v42, v46, v47" tail "\n"
#define T2_3(op, tail) op " v43, v43, v47, v44" tail "\n"
#define BITOP_BANKS_DISTINCT(n) X3_##n("v_bitop3_b32", " bitop3:0x96")
#define BITOP_BANKS_SAME(n) T2_##n("v_bitop3_b32", " bitop3:0x96")
#define BITOP_X4(n) S3_##n("v_bitop3_b32", " bitop3:0x96")
// Is the scalar operand a PORT (one SGPR-reading VALU instruction per four clocks, others issue meanwhile) or a
// property of the instruction (the SGPR reader occupies the VALU for four clocks)?  Mixes of one SGPR-reading
// v_xor_b32 with one and with three VGPR-only ones: a port gives max(4, 2 n) clocks per group, an occupied VALU 4 + 2 (n - 1).
#define XOR_MIX2(n) "v_xor_b32 %[a" #n "], %[sk], %[a" #n "]\n v_xor_b32 %[b" #n "], %[vk], %[b" #n "]\n"
#define XOR_MIX4(n) "v_xor_b32 %[a" #n "], %[sk], %[a" #n "]\n v_xor_b32 %[b" #n "], %[vk], %[b" #n "]\n v_add_u32 %[b" #n "], %[vk], %[b" #n "]\n v_xor_b32 %[b" #n "], %[vk], %[b" #n "]\n"
// ... and of a half-rate instruction with full-rate ones: do they share one issue pipe (4 + 2 n) or overlap (max)?
#define MAD_MIX2(n) "v_mad_u64_u32 %[w" #n "], vcc, %[a" #n "], %[vk], 0\n v_xor_b32 %[b" #n "], %[vk], %[b" #n "]\n"
#define MAD_MIX3(n) "v_mad_u64_u32 %[w" #n "], vcc, %[a" #n "], %[vk], 0\n v_xor_b32 %[b" #n "], %[vk], %[b" #n "]\n v_add_u32 %[b" #n "], %[vk], %[b" #n "]\n"
#define MADS_MIX2(n) "v_mad_u64_u32 %[w" #n "], vcc, %[a" #n "], %[sk], 0\n v_xor_b32 %[b" #n "], %[sk], %[b" #n "]\n"
// Does the ORDER of half-rate and full-rate instructions in a wave's stream matter?  64 instructions per block, half
// of them v_mad_u64_u32, half v_bitop3_b32 (three VGPRs), in runs of 1, 2, 4, 8 and 32; register sets rotate inside a run.
#define M_(n) "v_mad_u64_u32 %[w" #n "], vcc, %[a" #n "], %[vk], 0\n"
#define B_(n) "v_bitop3_b32 %[b" #n "], %[b" #n "], %[a" #n "], %[vk] bitop3:0x96\n"
#define F_(n) "v_fma_f32 %[f" #n "], %[f" #n "], %[vf], %[vg]\n"
#define ORD_RUN1 M_(0) B_(0) M_(1) B_(1) M_(2) B_(2) M_(3) B_(3)
#define ORD_RUN2 M_(0) M_(1) B_(0) B_(1) M_(2) M_(3) B_(2) B_(3)
#define ORD_RUN4 M_(0) M_(1) M_(2) M_(3) B_(0) B_(1) B_(2) B_(3)
#define ORD_RUN8 M_(0) M_(1) M_(2) M_(3) M_(0) M_(1) M_(2) M_(3) B_(0) B_(1) B_(2) B_(3) B_(0) B_(1) B_(2) B_(3)
#define ORD_M8 M_(0) M_(1) M_(2) M_(3) M_(0) M_(1) M_(2) M_(3)
#define ORD_B8 B_(0) B_(1) B_(2) B_(3) B_(0) B_(1) B_(2) B_(3)
#define ORD_MF1 M_(0) F_(0) M_(1) F_(1) M_(2) F_(2) M_(3) F_(3)
#define ORD_MF4 M_(0) M_(1) M_(2) M_(3) F_(0) F_(1) F_(2) F_(3)
#define ORD_BF1 B_(0) F_(0) B_(1) F_(1) B_(2) F_(2) B_(3) F_(3)
#define FMA_DISTINCT(n) X3_##n("v_fma_f32", "")
#define FMA_X4(n) S3_##n("v_fma_f32", "")
#define FMA_T2(n) T2_##n("v_fma_f32", "")
#define FMA_3V(n) "v_fma_f32 %[f" #n "], %[f" #n "], %[vf], %[vg]\n"
#define FMAC_2V(n) "v_fmac_f32 %[f" #n "], %[vf], %[vg]\n"
#define XOR_2V(n) "v_xor_b32 %[a" #n "], %[b" #n "], %[a" #n "]\n"
#define AND_L(n) "v_and_b32 %[a" #n "], 0x7ffff, %[a" #n "]\n"
#define OR_I(n) "v_or_b32 %[a" #n "], 1.0, %[a" #n "]\n"
#define LSHL_I(n) "v_lshlrev_b32 %[a" #n "], 2, %[a" #n "]\n"
#define SUB_V(n) "v_sub_u32 %[a" #n "], %[vk], %[a" #n "]\n"
#define MULF_E64_S(n) "v_mul_f32_e64 %[f" #n "], %[f" #n "], %[sf]\n"
#define MAX_V(n) "v_max_f32 %[f" #n "], %[vf], %[f" #n "]\n"
#define ADDF64(n) "v_add_f64 %[w" #n "], %[w" #n "], %[w" #n "]\n"
#define ADDCO(n) "v_add_co_u32 %[a" #n "], vcc, %[vk], %[a" #n "]\n"
#define CNDMASK_S(n) "v_cndmask_b32 %[a" #n "], %[a" #n "], %[b" #n "], s[10:11]\n"
#define READFIRST(n) "v_readfirstlane_b32 s1" #n ", %[a" #n "]\n"
#define DPP_ROW(n) "v_mov_b32_dpp %[a" #n "], %[b" #n "] row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define DS_SWZ(n) "ds_bpermute_b32 %[a" #n "], %[b" #n "], %[a" #n "]\n"

template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned *out, unsigned sk, float sf, unsigned long long *clk) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned a0 = threadIdx.x * 2654435761u + sk, a1 = a0 ^ 0x9E3779B9u, a2 = a0 + 12345u, a3 = a1 + 777u;
  unsigned b0 = a0 * 3u, b1 = a1 * 5u, b2 = a2 * 7u, b3 = a3 * 9u, vk = a0 | 1u;
  unsigned long long w0 = a0, w1 = a1, w2 = a2, w3 = a3;
  float f0 = 1.0f, f1 = 1.0001f, f2 = 0.5f, f3 = 1.5f, vf = 1.0000001f, vg = 1e-9f;
  asm volatile("" : "+v"(vk), "+v"(vf), "+v"(vg));
  for (int i = 0; i < kIters; ++i) {
    if constexpr (OP < 0) {
    }
    PROBE(0, XOR_V) PROBE(1, XOR_S) PROBE(2, XOR_L) PROBE(3, XOR_E64_S) PROBE(4, ADD_V) PROBE(5, ADD_S) PROBE(6, LSHR_I)
    PROBE(10, BITOP_V) PROBE(11, BITOP_S) PROBE(12, ANDOR_V) PROBE(13, ANDOR_S) PROBE(14, LSHLADD_V) PROBE(15, LSHLADD_S)
    PROBE(16, ADD3_V) PROBE(17, OR3_V) PROBE(18, XAD_V) PROBE(19, BFE_I) PROBE(20, BFI_V) PROBE(21, ALIGNBIT_V) PROBE(22, PERM_V)
    PROBE(23, SDWA_V) PROBE(24, SDWA_S)
    PROBE(30, MAD64_V) PROBE(31, MAD64_S) PROBE(32, MAD64_SGPRCARRY) PROBE(33, MAD64_ADD) PROBE(34, MULLO_V) PROBE(35, MULHI_V)
    PROBE(36, MULHI_S) PROBE(37, MAD24_V)
    PROBE(40, FMA_V) PROBE(41, FMA_S) PROBE(42, FMAC_V) PROBE(43, FMAC_L) PROBE(44, FMAMK) PROBE(45, MUL_V) PROBE(46, MUL_S)
    PROBE(47, MUL_L) PROBE(48, ADDF_V) PROBE(49, CVT_I32) PROBE(50, CVT_U32) PROBE(51, MOV_V) PROBE(52, CNDMASK) PROBE(53, CMP)
    PROBE(54, PKFMA) PROBE(55, PKMUL) PROBE(56, FMA64) PROBE(57, LSHL64)
    else if constexpr (OP == 90) { asm volatile(ORD_RUN1 ORD_RUN1 ORD_RUN1 ORD_RUN1 ORD_RUN1 ORD_RUN1 ORD_RUN1 ORD_RUN1 OPERANDS); }
    else if constexpr (OP == 91) { asm volatile(ORD_RUN2 ORD_RUN2 ORD_RUN2 ORD_RUN2 ORD_RUN2 ORD_RUN2 ORD_RUN2 ORD_RUN2 OPERANDS); }
    else if constexpr (OP == 92) { asm volatile(ORD_RUN4 ORD_RUN4 ORD_RUN4 ORD_RUN4 ORD_RUN4 ORD_RUN4 ORD_RUN4 ORD_RUN4 OPERANDS); }
    else if constexpr (OP == 93) { asm volatile(ORD_RUN8 ORD_RUN8 ORD_RUN8 ORD_RUN8 OPERANDS); }
    else if constexpr (OP == 94) { asm volatile(ORD_M8 ORD_M8 ORD_M8 ORD_M8 ORD_B8 ORD_B8 ORD_B8 ORD_B8 OPERANDS); }
    else if constexpr (OP == 95) { asm volatile(ORD_MF1 ORD_MF1 ORD_MF1 ORD_MF1 ORD_MF1 ORD_MF1 ORD_MF1 ORD_MF1 OPERANDS); }
    else if constexpr (OP == 96) { asm volatile(ORD_MF4 ORD_MF4 ORD_MF4 ORD_MF4 ORD_MF4 ORD_MF4 ORD_MF4 ORD_MF4 OPERANDS); }
    else if constexpr (OP == 97) { asm volatile(ORD_BF1 ORD_BF1 ORD_BF1 ORD_BF1 ORD_BF1 ORD_BF1 ORD_BF1 ORD_BF1 OPERANDS); }
    PROBE(60, BITOP_BANKS_DISTINCT) PROBE(61, BITOP_BANKS_SAME) PROBE(62, BITOP_X4) PROBE(64, FMA_DISTINCT)
    PROBE(65, FMA_X4) PROBE(80, FMA_T2) PROBE(81, XOR_MIX2) PROBE(82, XOR_MIX4) PROBE(83, MAD_MIX2) PROBE(84, MAD_MIX3) PROBE(85, MADS_MIX2) PROBE(66, FMA_3V) PROBE(67, FMAC_2V) PROBE(68, XOR_2V) PROBE(69, AND_L) PROBE(70, OR_I) PROBE(71, LSHL_I)
    PROBE(72, SUB_V) PROBE(73, MULF_E64_S) PROBE(74, MAX_V) PROBE(75, ADDF64) PROBE(76, ADDCO) PROBE(77, CNDMASK_S) PROBE(78, READFIRST)
    PROBE(79, DPP_ROW)
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ b0 ^ b1 ^ b2 ^ b3 ^ (unsigned)w0 ^ (unsigned)w1 ^ (unsigned)w2 ^
                                               (unsigned)w3 ^ __float_as_uint(f0 + f1 + f2 + f3);
}

template <int OP>
int run(const char *name, const char *operands, unsigned *d_out, int cus, unsigned long long *d_clk) {
  for (int waves_per_simd : {4, 8}) {
    const int grid = cus * waves_per_simd;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, d_out, 3u, 1.5f, d_clk);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(probe<OP>, dim3(grid), dim3(256), 0, 0, d_out, 5u, 1.5f, d_clk);
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
    const double per_simd_per_s = (double)kIters * kPerIter * waves_per_simd / (ms * 1e-3);
    printf("{\"probe\": \"%s\", \"operands\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ginst_per_s_per_simd\": %.4f, "
           "\"clock_ghz\": %.3f, \"clk_per_inst\": %.3f}\n", name, operands, waves_per_simd, ms, per_simd_per_s * 1e-9, ghz,
           ghz / (per_simd_per_s * 1e-9));
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
#define RUN(ID, NAME, OPS) if (run<ID>(NAME, OPS, d_out, cus, d_clk)) return 1;
  RUN(0, "v_xor_b32", "vgpr") RUN(1, "v_xor_b32", "sgpr") RUN(2, "v_xor_b32", "literal") RUN(3, "v_xor_b32_e64", "sgpr")
  RUN(4, "v_add_u32", "vgpr") RUN(5, "v_add_u32", "sgpr") RUN(6, "v_lshrrev_b32", "inline")
  RUN(10, "v_bitop3_b32", "vgpr") RUN(11, "v_bitop3_b32", "sgpr") RUN(12, "v_and_or_b32", "vgpr") RUN(13, "v_and_or_b32", "sgpr+inline")
  RUN(14, "v_lshl_add_u32", "vgpr") RUN(15, "v_lshl_add_u32", "sgpr") RUN(16, "v_add3_u32", "vgpr") RUN(17, "v_or3_b32", "vgpr")
  RUN(18, "v_xad_u32", "vgpr") RUN(19, "v_bfe_u32", "inline") RUN(20, "v_bfi_b32", "vgpr") RUN(21, "v_alignbit_b32", "vgpr+inline")
  RUN(22, "v_perm_b32", "vgpr") RUN(23, "v_and_b32_sdwa", "vgpr") RUN(24, "v_and_b32_sdwa", "sgpr")
  RUN(30, "v_mad_u64_u32", "vgpr, vcc carry") RUN(31, "v_mad_u64_u32", "sgpr, vcc carry") RUN(32, "v_mad_u64_u32", "sgpr, sgpr-pair carry")
  RUN(33, "v_mad_u64_u32", "vgpr, 64-bit addend") RUN(34, "v_mul_lo_u32", "vgpr") RUN(35, "v_mul_hi_u32", "vgpr") RUN(36, "v_mul_hi_u32", "sgpr")
  RUN(37, "v_mad_u32_u24", "vgpr")
  RUN(40, "v_fma_f32", "vgpr") RUN(41, "v_fma_f32", "sgpr") RUN(42, "v_fmac_f32", "vgpr") RUN(43, "v_fmac_f32", "literal")
  RUN(44, "v_fmamk_f32", "literal") RUN(45, "v_mul_f32", "vgpr") RUN(46, "v_mul_f32", "sgpr") RUN(47, "v_mul_f32", "literal")
  RUN(48, "v_add_f32", "vgpr") RUN(49, "v_cvt_f32_i32", "vgpr") RUN(50, "v_cvt_f32_u32", "vgpr") RUN(51, "v_mov_b32", "vgpr")
  RUN(52, "v_cndmask_b32", "vgpr, vcc") RUN(53, "v_cmp_lt_u32", "vgpr -> vcc") RUN(54, "v_pk_fma_f32", "vgpr") RUN(55, "v_pk_mul_f32", "vgpr")
  RUN(56, "v_fma_f64", "vgpr") RUN(57, "v_lshlrev_b64", "inline")
  RUN(60, "v_bitop3_b32", "explicit registers, three banks") RUN(61, "v_bitop3_b32", "explicit registers, two sources in one bank")
  RUN(62, "v_bitop3_b32", "explicit registers, all sources in one bank") 
  RUN(64, "v_fma_f32", "explicit registers, three banks") RUN(65, "v_fma_f32", "explicit registers, all sources in one bank")
  RUN(80, "v_fma_f32", "explicit registers, two sources in one bank")
  RUN(90, "order: 32 v_mad_u64_u32 + 32 v_bitop3_b32", "runs of 1 (alternating)") RUN(91, "order: 32 v_mad_u64_u32 + 32 v_bitop3_b32", "runs of 2")
  RUN(92, "order: 32 v_mad_u64_u32 + 32 v_bitop3_b32", "runs of 4") RUN(93, "order: 32 v_mad_u64_u32 + 32 v_bitop3_b32", "runs of 8")
  RUN(94, "order: 32 v_mad_u64_u32 + 32 v_bitop3_b32", "runs of 32") RUN(95, "order: 32 v_mad_u64_u32 + 32 v_fma_f32", "runs of 1 (alternating)")
  RUN(96, "order: 32 v_mad_u64_u32 + 32 v_fma_f32", "runs of 4") RUN(97, "order: 32 v_bitop3_b32 + 32 v_fma_f32", "runs of 1 (alternating)")
  RUN(81, "mix: v_xor sgpr + v_xor vgpr", "per GROUP of 2: count x 2") RUN(82, "mix: v_xor sgpr + 3 vgpr-only", "per GROUP of 4: count x 4")
  RUN(83, "mix: v_mad_u64_u32 + v_xor vgpr", "per GROUP of 2: count x 2") RUN(84, "mix: v_mad_u64_u32 + 2 vgpr-only", "per GROUP of 3: count x 3")
  RUN(85, "mix: v_mad_u64_u32 sgpr + v_xor sgpr", "per GROUP of 2: count x 2")
  RUN(66, "v_fma_f32", "three distinct vgprs") RUN(67, "v_fmac_f32", "two distinct vgprs + dst") RUN(68, "v_xor_b32", "two vgprs")
  RUN(69, "v_and_b32", "literal") RUN(70, "v_or_b32", "inline 1.0") RUN(71, "v_lshlrev_b32", "inline") RUN(72, "v_sub_u32", "vgpr")
  RUN(73, "v_mul_f32_e64", "sgpr") RUN(74, "v_max_f32", "vgpr") RUN(75, "v_add_f64", "vgpr") RUN(76, "v_add_co_u32", "vgpr -> vcc")
  RUN(77, "v_cndmask_b32", "vgpr, sgpr-pair mask") RUN(78, "v_readfirstlane_b32", "vgpr -> sgpr") RUN(79, "v_mov_b32_dpp", "row_shr:1")
  return 0;
}
