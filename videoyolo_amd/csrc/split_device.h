// split_device.h — device helpers shared by the split-fp32 kernels (conv_split.hip, conv_wino.hip): the bf16 x 3 cut of
// eight fp32 channels and the LDS slot swizzle of their tile images.
#pragma once

#include "conv_device.h"

// MFMA shape.  1: v_mfma_f32_16x16x32_bf16 — its 32 k-slots take TWO partial products of a 16-channel k-step (lanes
// 0-31 feed one pair of planes, lanes 32-63 another), three instructions per 16x16 tile; 0: v_mfma_f32_32x32x16_bf16,
// six per 32x32 tile.  Same FLOPs per cycle; the chip holds a higher clock on the 16x16 shape (MI355X_MICROARCH.md,
// DVFS give-back item 7), and this kernel is bound by the power budget (all-zero operands: 275 instead of 212 TF).
#ifndef VY_SPLIT_M16
#define VY_SPLIT_M16 0
#endif
// LDS 16-B slot of channel octet `oct` in row `row`: the 32x32x16 fragment read (32 rows x one octet per half-wave)
// needs the XOR with bit 3 of the row to be conflict-free; the 16x16x32 read (16 rows x two octets) is conflict-free
// on the plain image and 2-way on the swizzled one
#define VY_SPLIT_SLOT(row, oct) (VY_SPLIT_M16 ? (oct) : ((oct) ^ (((row) >> 3) & 1)))

#if defined(__HIP_DEVICE_COMPILE__)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// two fp32 -> one dword of two bf16 (RNE), element 0 in the low half
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

typedef float vy_f32x2 __attribute__((ext_vector_type(2)));

// 8 consecutive channels -> the three bf16 planes (4 dwords each).  The two differences of a channel pair are written
// as 2-vectors: hipcc issues them as one v_pk_add_f32 (negated operand) — 9 instead of 11 vector instructions per pair,
// same IEEE results
__device__ __forceinline__ void split8(const f32x4 v0, const f32x4 v1, vy_u32x4& H, vy_u32x4& M, vy_u32x4& L) {
  const vy_f32x2 x[4] = {{v0[0], v0[1]}, {v0[2], v0[3]}, {v1[0], v1[1]}, {v1[2], v1[3]}};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned h = cvt_pk_bf16(x[j][0], x[j][1]);
    const vy_f32x2 hf = {__builtin_bit_cast(float, h << 16), __builtin_bit_cast(float, h & 0xffff0000u)};
    const vy_f32x2 r = x[j] - hf;
    const unsigned m = cvt_pk_bf16(r[0], r[1]);
    const vy_f32x2 mf = {__builtin_bit_cast(float, m << 16), __builtin_bit_cast(float, m & 0xffff0000u)};
    const vy_f32x2 l = r - mf;
    H[j] = h;
    M[j] = m;
    L[j] = cvt_pk_bf16(l[0], l[1]);
  }
}
#endif

