// smmc_device.h -- device helpers shared by the kernel translation units (not installed).
//
// The compounding step (src/simulations.cpp:14-16 of the reference), the proven divide-by-100
// shortcut, the wave reductions and two gfx950 instruction wrappers.  Everything here is
// __forceinline__; both smmc_kernels.hip and smmc_ref_kernels.hip must give the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include "smmc_internal.h"

namespace smmc {
namespace dev {

// a ^ b ^ c in one VALU instruction (gfx950 v_bitop3_b32, truth table 0x96); hipcc
// does not form it from two chained XORs on its own.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// x / 100.0f, correctly rounded, in TWO instructions: 1/100 = ch + cl with ch = fl(1/100) and
// cl = fl(1/100 - ch); fma(x, ch, fl(x cl)) then equals the IEEE quotient for every binary32 x with
// |x| >= 2^-114 (Brisebarre and Muller's multiplication by a constant held in two words; whether it is
// exact for a given constant has to be checked, and for 1/100 it is: exhaustive CPU proof over all
// 2^23 mantissas of every exponent in tests/test_numerics_cpu.py, device self-test over both signs).
// Round 1's form -- q = x ch, one FMA for the residual, one for the correction -- took three.  The
// host only selects it when every product total * a provably stays inside [2^-89, 2^127): there
// x cl is a normal number as well.
template <bool kExactDiv>
__device__ __forceinline__ float div100(float m) {
  if constexpr (kExactDiv) {
    return m / 100.0f;  // IEEE divide (hipcc default: correctly rounded)
  } else {
    const float ch = 0.01f, cl = 0x1.eb851ep-33f;
    return __builtin_fmaf(m, ch, m * cl);
  }
}

// One compounding period, src/simulations.cpp:14-16 with a = 100.0f + r formed
// by the caller: total * a, then / 100.
template <bool kExactDiv>
__device__ __forceinline__ float compound(float total, float a) {
  const float m = total * a;
  return div100<kExactDiv>(m);
}

// A load from an LDS byte address held in a register.  The v3 tables sit at LDS address 0 (the
// Gaussian kernels have no static __shared__ data -- the host checks that, static_lds_bytes() -- and
// stage them first in the dynamic allocation), so a bin's byte offset IS its address: going through
// the `extern __shared__` symbol instead costs a v_add with a link-time constant that turns out to be 0.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ T lds_load_at(uint32_t byte_addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) T *>(static_cast<uintptr_t>(byte_addr));
}

// A wave-uniform constant as a VGPR operand.  In a stream of its own a VALU instruction with an SGPR source issues
// at half rate on gfx950 (v_bitop3_b32 2.3 clk per wave with three VGPRs, 4.1 with a scalar; tools/ubench_ops.hip);
// in the kernels' mixes the scalar operand is a port that is rarely full, and holding the constants of the
// three-input logic operations in VGPRs (VOP3 takes no literals; the compiler puts every uniform value into an SGPR)
// is worth 0.2-0.8 % (profiles/r04/ab_operands.txt, ab_table.txt) -- kept because it is free.
__device__ __forceinline__ uint32_t vgpr_const(uint32_t c) {
  asm("" : "+v"(c));
  return c;
}

// kDiv: how total * a / 100 is formed.  kDivFast: the reciprocal-multiply form, the host has proven
// that no product of any path leaves its domain.  kDivExact: the IEEE divide.  kDivChecked: the
// fast form while the path stays inside a window [chk_lo, chk_hi] tested once per Philox block --
// wide enough that no product of the FOLLOWING block can leave the domain, whatever it draws --
// and, for a lane that ever leaves it, the whole path again with the IEEE divide (a real returns
// table with a +42 % month cannot be proven safe for 360 periods, yet no path ever gets there).
enum : int { kDivFast = 0, kDivExact = 1, kDivChecked = 2 };

// ---- reductions: wave shuffles, then LDS across the 4 waves ------------------

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
  return v;
}

constexpr int kWaves = kBlock / 64;

}  // namespace dev
}  // namespace smmc
