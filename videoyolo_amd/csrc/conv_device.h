// conv_device.h — device-side helpers shared by the implicit-GEMM conv kernels (conv_igemm.hip: 32x32x2 MFMA tiles,
// conv_small.hip: 16x16x4 MFMA tiles for launches that cannot fill the chip with 32x32 wave tiles).
#pragma once
#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// One LDS-DMA instruction (global_load_lds_dwordx4: 64 lanes x 16 B -> 1 KiB at lds_base + lane*16), issued
// as inline asm so that hipcc does not serialise it against the surrounding ds_reads (it would wait
// vmcnt(0) before every LDS read that follows a DMA it knows about).  Ordering is by hand: every wave
// executes `s_waitcnt vmcnt(0)` before the barrier that precedes the first read of the tile.
__device__ __forceinline__ void lds_dma16(const float* gptr, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory", "m0");
#endif
}

// The same with the address split into a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset:
// the base advances per k-step with two scalar adds, the per-lane offsets never change inside the k-loop — no
// vector instruction per DMA (the fp32 MFMA shares the vector pipe's FMA hardware: every VALU instruction in the
// loop is paid in matrix time, tools/probe/coissue_probe.hip).
__device__ __forceinline__ void lds_dma16_s(unsigned voff, const float* sbase, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory", "m0");
#endif
}

// fp32 access through a buffer descriptor (the raw_buffer builtins move 32-bit integers): byte offset
// `voff` per lane + uniform `soff`; an offset past the descriptor's range reads 0 / is not written
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0));
}
__device__ __forceinline__ void buf_store_f32(float v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff, soff, 0);
}
// 16-byte write-through store / L1-bypassing load (aux 16 = sc1): the pair that hands bytes to another workgroup inside a
// launch without an agent-scope release (an L2 write-back of the whole XCD) or acquire — MI355X_MICROARCH.md, visibility
typedef unsigned vy_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void buf_store_f32x4_sc1(f32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vy_u32x4, v), rsrc, voff, 0, 16);
}
__device__ __forceinline__ f32x4 buf_load_f32x4_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 16));
}
// the same with the default cache policy (write-through L1, write-back L2): bytes the SAME lanes read back later
__device__ __forceinline__ void buf_store_f32x4(f32x4 v, __amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vy_u32x4, v), rsrc, voff, 0, 0);
}
__device__ __forceinline__ f32x4 buf_load_f32x4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
}
#endif

__device__ __forceinline__ unsigned fd_div(unsigned n, const VyFastDiv f) {
  const unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

// workgroup barrier that waits for this wave's LDS traffic only (not for outstanding global loads: the deep
// pipeline below keeps LDS-DMA of later k-steps in flight across it)
__device__ __forceinline__ void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

