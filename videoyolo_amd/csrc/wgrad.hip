// wgrad.hip — weight gradient of the 3x3 / 1x1 convolutions on the fp32 matrix core.
//
// What mxnet's autograd computes for Convolution's weight under train_yolov3.py:631:
//   dW[o][tap][cin] = sum over output pixels p of  dz[p][o] * a[p*stride + tap][cin]
// GEMM view: D[o][n] = sum_p A[o][p] * B[p][n],  n = tap*Cin + cin; the reduction runs over the
// B*Ho*Wo pixels (10^4..10^6) while the output is small, so the pixel range is split over blocks
// (split-K); each split writes its own fp32 slab and vy_launch_slab_reduce adds the slabs in index
// order (deterministic, no atomics).
//
// Both operands are "k-major" in HBM already: a pixel's channel vector is contiguous, so the LDS
// tiles are [32 pixels][128 channels] exactly as loaded (LDS-DMA, 16 B per lane, rows of 512 B) and
// the MFMA fragments A[i=o][k=p], B[k=p][j=n] are ds_read_b32 of 32 consecutive floats per
// half-wave (conflict-free, no swizzle).  Block tile 128(o) x 128(n) x 32(p), 4 waves 2x2,
// 64 accumulator VGPRs per lane, 64 KiB LDS double buffer -> 2 blocks / CU.
#include <cstdlib>

#include "kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// LDS-DMA as inline asm (see conv_igemm.hip: keeps hipcc from serialising it against the ds_reads);
// ordered by hand: s_waitcnt vmcnt(0) before the barrier that precedes the first read of the tile.
__device__ __forceinline__ void lds_dma16(const float* gptr, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory", "m0");
#endif
}

// KP = pixels per k-step (LDS stage): 32 -> 64 KiB double buffer, 2 blocks / CU.  (16 -> 32 KiB and 4 blocks / CU
// was measured: faster alone, slower beside the dgrad chain it shares the chip with — 407 vs 414 frames/s.)
template <int KP>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 128, TILE = KP * 128 * 4, STAGE = 2 * TILE;
  constexpr int NJ = KP / 8;  // DMA instructions per wave per operand per k-step = MFMA groups per k-step
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, lrow = lane & 31;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int o0 = tile_m * BM, n0 = tile_n * BN;
  const int Ntot = a.k * a.k * a.Cin;
  const int split = blockIdx.y;
  const int p_begin = split * a.k_per_split;
  int p_end = p_begin + a.k_per_split;
  if (p_end > a.M) p_end = a.M;
  const int T = (p_end - p_begin + KP - 1) / KP;

  // this lane's fixed column chunk in both tiles
  const int chunk = lane & 31;            // 16-B chunk inside a 512-B row
  const int ao = o0 + chunk * 4;          // dz channel of this chunk
  const bool a_ok = ao < a.z_cs;          // (padded) channel exists in the dz plane
  const int bn = n0 + chunk * 4;          // n column of this chunk
  const bool b_ok = bn < Ntot;
  const int tap = b_ok ? bn / a.Cin : 0;
  const int cin = b_ok ? bn - tap * a.Cin : 0;
  const int pad = a.k >> 1;
  const int dy = a.k == 3 ? tap / 3 - pad : 0, dx = a.k == 3 ? tap % 3 - pad : 0;
  const int Hzp = a.Ho + 2, Wzp = a.Wo + 2;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Each lane stages the same NJ pixel rows of both tiles every k-step (row = 2*(j*4+wave) + h).  Their
  // source pointers are carried incrementally: +KP pixels per k-step, plus constant skips over the zero
  // border when the pixel index wraps to the next image row / the next image — no integer division and
  // no 64-bit multiplies inside the k-loop.
  int py[NJ], px[NJ];
  const float* zp[NJ];
  const float* ap[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int p = p_begin + 2 * (j * 4 + wave) + h;
    px[j] = p % a.Wo;
    const int tt = p / a.Wo;
    py[j] = tt % a.Ho;
    const int b = tt / a.Ho;
    zp[j] = a.dz + ((long long)(b * Hzp + py[j] + 1) * Wzp + px[j] + 1) * a.z_cs + ao;
    ap[j] = a.a + ((long long)(b * a.a_Hp + py[j] * a.stride + 1 + dy) * a.a_Wp + px[j] * a.stride + 1 + dx) * a.a_cs +
            a.a_co + cin;
  }
  const long long z_step = (long long)KP * a.z_cs, a_step = (long long)KP * a.stride * a.a_cs;
  const long long z_row = 2LL * a.z_cs, a_row = (long long)(a.a_Wp - a.Wo) * a.stride * a.a_cs;
  const long long z_img = 2LL * Wzp * a.z_cs, a_img = (long long)(a.a_Hp - a.Ho * a.stride) * a.a_Wp * a.a_cs;
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  // one quarter (instruction j) of the LDS-DMA of tile t into buffer buf
  auto stage = [&](int t, int buf, int j) {
    const int q = j * 4 + wave;
    const bool live = p_begin + t * KP + 2 * q + h < p_end;
    lds_dma16(live && a_ok ? zp[j] : a.zero, lds0 + buf * STAGE + q * 1024);
    lds_dma16(live && b_ok ? ap[j] : a.zero, lds0 + buf * STAGE + TILE + q * 1024);
    // advance this row by KP pixels for the next tile
    px[j] += KP;
    zp[j] += z_step;
    ap[j] += a_step;
    while (px[j] >= a.Wo) {
      px[j] -= a.Wo;
      zp[j] += z_row;
      ap[j] += a_row;
      if (++py[j] == a.Ho) {
        py[j] = 0;
        zp[j] += z_img;
        ap[j] += a_img;
      }
    }
  };

  if (T > 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) stage(0, 0, j);
  }
  for (int t = 0; t < T; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of tile t has landed
    __syncthreads();
    const bool more = t + 1 < T;
    const float* tA = reinterpret_cast<const float*>(smem + (t & 1) * STAGE);
    const float* tB = tA + KP * 128;
#pragma unroll
    for (int g = 0; g < NJ; ++g) {
      // one ds_read_b64 per operand and pixel pair: lane lrow gets rows / columns 2*lrow and 2*lrow+1 of the
      // wave's 64, so MFMA tile m of an operand covers the interleaved set {2*l + m} (undone in the epilogue)
      f32x2 av[4], bv[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int krow = (2 * (g * 4 + s4) + h) * 128;
        av[s4] = *reinterpret_cast<const f32x2*>(tA + krow + wm * 64 + 2 * lrow);
        bv[s4] = *reinterpret_cast<const f32x2*>(tB + krow + wn * 64 + 2 * lrow);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s4][i], bv[s4][j], acc[i][j], 0, 0, 0);
        // a quarter of the next tile's DMA, issued while this group's MFMAs occupy the matrix pipe
        if (s4 == 0 && more) stage(t + 1, (t + 1) & 1, g);
      }
    }
  }

  // acc[i][j][r] = D[o = o0 + wm*64 + 2*row(r,h) + i][n = n0 + wn*64 + 2*lrow + j]: the two column tiles of
  // a lane are adjacent in memory -> 8-B stores, 256 B contiguous per half-wave
  float* slab = a.slabs + (long long)split * a.Cout * Ntot;
  const int n = n0 + wn * 64 + 2 * lrow;  // even; Ntot is a multiple of 32
  if (n < Ntot) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + i;
        if (o < a.Cout) {
          f32x2 v = {acc[i][0][r], acc[i][1][r]};
          *reinterpret_cast<f32x2*>(slab + (long long)o * Ntot + n) = v;
        }
      }
  }
#endif
}

hipError_t vy_launch_wgrad(const WgradArgs& a, hipStream_t s) {
  if (a.Cin % 32 != 0 || a.k_per_split % 32 != 0 || a.splits < 1 || (a.z_cs & 3) || (a.a_cs & 3) || (a.a_co & 3))
    return hipErrorInvalidValue;
  const int Ntot = a.k * a.k * a.Cin;
  const int tiles_m = (a.Cout + 127) / 128, tiles_n = (Ntot + 127) / 128;
  hipLaunchKernelGGL(wgrad_kernel<32>, dim3(tiles_m * tiles_n, a.splits), dim3(256), 0, s, a, tiles_n);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int splits, long long n,
                                                          float* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.0f;
  for (int k = 0; k < splits; ++k) s += slabs[(long long)k * n + i];
  dst[i] = s;
}

// Many slabs over a small weight tensor (the early layers: up to ~500 slabs of a few thousand floats):
// a thread per element would walk the slabs serially, so G thread groups each add every G-th slab of 32
// columns and the G partial sums are combined in group order (still a fixed order: deterministic).
template <int G>
__global__ __launch_bounds__(32 * G) void slab_reduce_wide_kernel(const float* __restrict__ slabs, int splits, long long n,
                                                                  float* __restrict__ dst) {
  __shared__ float part[G][33];
  const int col = threadIdx.x & 31, g = threadIdx.x >> 5;
  const long long i = (long long)blockIdx.x * 32 + col;
  float s0 = 0.0f, s1 = 0.0f;
  if (i < n) {
    int k = g;
    for (; k + G < splits; k += 2 * G) {
      s0 += slabs[(long long)k * n + i];
      s1 += slabs[(long long)(k + G) * n + i];
    }
    if (k < splits) s0 += slabs[(long long)k * n + i];
  }
  part[g][col] = s0 + s1;
  __syncthreads();
  if (g == 0 && i < n) {
    float s = 0.0f;
#pragma unroll
    for (int q = 0; q < G; ++q) s += part[q][col];
    dst[i] = s;
  }
}

hipError_t vy_launch_slab_reduce(const float* slabs, int splits, long long n, float* dst, hipStream_t s) {
  if (splits >= 64)
    hipLaunchKernelGGL(slab_reduce_wide_kernel<32>, dim3((unsigned)((n + 31) / 32)), dim3(1024), 0, s, slabs, splits, n, dst);
  else if (splits >= 12)
    hipLaunchKernelGGL(slab_reduce_wide_kernel<8>, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, slabs, splits, n, dst);
  else
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, slabs, splits, n, dst);
  return hipGetLastError();
}
