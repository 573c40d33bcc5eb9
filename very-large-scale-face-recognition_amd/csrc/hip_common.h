// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common_host.h"

namespace vlsfr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16;

#define VLSFR_LDS __attribute__((address_space(3)))

// D(16x16 f32) += A(16x32 bf16) * B(32x16 bf16).  Lane l = 16*h + r holds
//   A[row r][k = 8h + j], B[k = 8h + j][col r]  (j = 0..7)   and   D[row 4h + e][col r] (e = 0..3).
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// The same MFMA with the accumulator pinned in the accumulator half of the register file ("+a") and the instruction's place in the
// stream fixed (asm volatile): for kernels whose accumulators (224 - 256 registers per wave) only fit there — left to the compiler
// they are shuttled between the two halves every iteration or spilled.  The compiler's hazard recogniser does not see an MFMA in
// it: a reader of `c` other than the next MFMA on it must be kept away by s_nops (conv_igemm_hw4_kernel's epilogue).
__device__ __forceinline__ void mfma16_agpr(f32x4& c, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// ... and with the accumulator in architectural VGPRs ("+v"): the small product of a kernel whose accumulator file is full
__device__ __forceinline__ void mfma16_vgpr(f32x4& c, bf16x8 a, bf16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements is delivered
// column-major — lane i of the group receives column i, rows 0..3.  Lane 4q + p of the group
// supplies the LDS address of row q, columns 4p..4p+3 (8-byte aligned).
__device__ __forceinline__ short4v lds_read_tr16(void* lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((VLSFR_LDS short4v*)lds_addr);
}

// Sum over the 16 lanes of a DPP row (lanes 16h .. 16h + 15), result in every lane of the row:
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror -- four VALU adds, no LDS.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// Sixteen per-lane values, each to be summed over the 16 lanes of a DPP row: lane r16 = k of the row gets the row sum of v[k].
// A transposing butterfly — in each of the four exchanges (row_mirror, row_half_mirror, quad_perm [2,3,0,1], quad_perm [1,0,3,2]:
// the partner differs in bit 3 / 2 / 1 / 0 of r16) a lane keeps the half of its values whose index has ITS bit and adds the
// partner's copy of them: 8 + 4 + 2 + 1 adds with two selects each (45 VALU) instead of sixteen 4-add row sums and 16 selects.
__device__ __forceinline__ float row16_fold16(const float (&v)[16], int r16) {
  const bool b3 = r16 & 8, b2 = r16 & 4, b1 = r16 & 2, b0 = r16 & 1;
  float w[8], x[4], y[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) w[k] = (b3 ? v[k + 8] : v[k]) + dpp_mov<0x140>(b3 ? v[k] : v[k + 8]);
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = (b2 ? w[k + 4] : w[k]) + dpp_mov<0x141>(b2 ? w[k] : w[k + 4]);
#pragma unroll
  for (int k = 0; k < 2; ++k) y[k] = (b1 ? x[k + 2] : x[k]) + dpp_mov<0x4E>(b1 ? x[k] : x[k + 2]);
  return (b0 ? y[1] : y[0]) + dpp_mov<0xB1>(b0 ? y[0] : y[1]);
}

// v + the value of lane (lane ^ O), without LDS traffic: v_permlane32_swap / v_permlane16_swap across the
// 16-lane rows (swap(v, v) returns (own, partner) in one half and (partner, own) in the other, so the sum of
// the pair is own + partner in every lane), DPP row rotations inside a row.  For O < 16 the rotation adds
// lane (lane + O) mod 16 -- used as a reduction butterfly (O = 8, 4, 2, 1 in that order) it yields the same
// group sums as the xor butterfly.
// (own, partner) of lane ^ 16 / lane ^ 32.  The swap instruction exchanges halves of TWO registers in place;
// written as inline asm on two read-write operands because the builtin, fed the same value twice, came back
// as r[0] + r[0] (scripts/probes/lane_steps.hip).  The s_nops cover the VALU-write -> permlane-read hazard the
// compiler cannot see inside the asm.
template <int O>
__device__ __forceinline__ void lane_pair(float v, float& r0, float& r1) {
  float x = v, y = v;
  if constexpr (O == 32) asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  else asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  r0 = x;
  r1 = y;
}
template <int O>
__device__ __forceinline__ float lane_step_sum(float v) {
  if constexpr (O >= 16) {
    float r0, r1;
    lane_pair<O>(v, r0, r1);
    return r0 + r1;
  } else {
    return v + dpp_mov<0x120 + O>(v);   // row_ror:O
  }
}
template <int O>
__device__ __forceinline__ float lane_step_max(float v) {
  if constexpr (O >= 16) {
    float r0, r1;
    lane_pair<O>(v, r0, r1);
    return fmaxf(r0, r1);
  } else {
    return fmaxf(v, dpp_mov<0x120 + O>(v));
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Per-launch timing of a kernel family while vlsfr_profile_enable(1) is in effect (include/vlsfr.h section 9):
// HIP events on the launch stream before and after the scope.  `work` = algorithmic FLOPs of the launch.
struct ProfScope {
  hipStream_t st;
  bool on;
  hipEvent_t a, b;
  int family;
  double work;
  ProfScope(hipStream_t s, int family, double work);
  ~ProfScope();
};

inline int hip_fail(hipError_t e, const char* what) {
  return fail(VLSFR_EHIP, "%s: %s", what, hipGetErrorString(e));
}

#define VLSFR_HIP_CHECK_LAUNCH(what)                              \
  do {                                                            \
    hipError_t e__ = hipGetLastError();                           \
    if (e__ != hipSuccess) return vlsfr::hip_fail(e__, what);     \
  } while (0)

}  // namespace vlsfr
