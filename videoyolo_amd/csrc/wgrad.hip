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
// BM = output channels per block tile: 128, or 64 for the layers with Cout <= 64 (the 208x208 / 104x104 convs of
// stage 0: on the 128-row tile half — for Cout = 32 three quarters — of every MFMA multiplied zero padding; they
// ran at 50 / 16 TFLOP/s).  With BM = 64 all four waves share the 64 dz channels and split the 128 n columns
// (wave tile 64 x 32, two accumulators), the dz stage is [KP][64] (256-B rows, four pixels per DMA instruction).
template <int KP, int BM>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(BM == 128 || BM == 64, "tile height");
  constexpr int BN = 128, TILE_A = KP * BM * 4, TILE_B = KP * 128 * 4, STAGE = TILE_A + TILE_B;
  constexpr int NJ = KP / 8;          // MFMA groups per k-step = DMA instructions per wave for a 128-column tile
  constexpr int NJA = NJ * BM / 128;  // DMA instructions per wave for the dz tile
  constexpr int TJ = BM == 128 ? 2 : 1;  // accumulator tiles along n per wave
  constexpr int A_CPR = BM / 4;       // 16-B chunks per dz row (32 or 16)
  constexpr int A_RPI = 64 / A_CPR;   // dz pixel rows per DMA instruction (2 or 4)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = BM == 128 ? wave >> 1 : 0, wn = BM == 128 ? wave & 1 : wave;
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
  const int a_chunk = lane % A_CPR, a_sub = lane / A_CPR;  // dz: chunk inside the row, pixel row inside the instruction
  const int ao = o0 + a_chunk * 4;        // dz channel of this chunk
  const bool a_ok = ao < a.z_cs;          // (padded) channel exists in the dz plane
  const int chunk = lane & 31;            // activation tile: 16-B chunk inside a 512-B row
  const int bn = n0 + chunk * 4;          // n column of this chunk
  const bool b_ok = bn < Ntot;
  const int tap = b_ok ? bn / a.Cin : 0;
  const int cin = b_ok ? bn - tap * a.Cin : 0;
  const int pad = a.k >> 1;
  const int dy = a.k == 3 ? tap / 3 - pad : 0, dx = a.k == 3 ? tap % 3 - pad : 0;
  const int Hzp = a.Ho + 2, Wzp = a.Wo + 2;

  f32x16 acc[2][TJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Each lane stages the same pixel rows of both tiles every k-step (activations: row 2*(j*4+wave) + h; dz:
  // row A_RPI*(j*4+wave) + a_sub).  Their source pointers are carried incrementally: +KP pixels per k-step,
  // plus constant skips over the zero border when the pixel index wraps to the next image row / the next
  // image — no integer division and no 64-bit multiplies inside the k-loop.
  int zy[NJA], zx[NJA], py[NJ], px[NJ];
  const float* zp[NJA];
  const float* ap[NJ];
#pragma unroll
  for (int j = 0; j < NJA; ++j) {
    const int p = p_begin + A_RPI * (j * 4 + wave) + a_sub;
    zx[j] = p % a.Wo;
    const int tt = p / a.Wo;
    zy[j] = tt % a.Ho;
    const int b = tt / a.Ho;
    zp[j] = a.dz + ((long long)(b * Hzp + zy[j] + 1) * Wzp + zx[j] + 1) * a.z_cs + ao;
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int p = p_begin + 2 * (j * 4 + wave) + h;
    px[j] = p % a.Wo;
    const int tt = p / a.Wo;
    py[j] = tt % a.Ho;
    const int b = tt / a.Ho;
    ap[j] = a.a + ((long long)(b * a.a_Hp + py[j] * a.stride + 1 + dy) * a.a_Wp + px[j] * a.stride + 1 + dx) * a.a_cs +
            a.a_co + cin;
  }
  const long long z_step = (long long)KP * a.z_cs, a_step = (long long)KP * a.stride * a.a_cs;
  const long long z_row = 2LL * a.z_cs, a_row = (long long)(a.a_Wp - a.Wo) * a.stride * a.a_cs;
  const long long z_img = 2LL * Wzp * a.z_cs, a_img = (long long)(a.a_Hp - a.Ho * a.stride) * a.a_Wp * a.a_cs;
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  // DMA of tile t into buffer buf: the activation instruction j of this wave, and (j < NJA) the dz instruction j
  auto stage = [&](int t, int buf, int j) {
    const int q = j * 4 + wave;
    const bool live = p_begin + t * KP + 2 * q + h < p_end;
    if constexpr (BM == 128) {
      // both tiles stage the same pixel rows: one set of coordinates advances both pointers
      lds_dma16(live && a_ok ? zp[j] : a.zero, lds0 + buf * STAGE + q * 1024);
      lds_dma16(live && b_ok ? ap[j] : a.zero, lds0 + buf * STAGE + TILE_A + q * 1024);
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
    } else {
      if (j < NJA) {  // dz: four pixel rows per instruction, its own coordinates
        const int jz = j < NJA ? j : 0;
        const bool zlive = p_begin + t * KP + A_RPI * q + a_sub < p_end;
        lds_dma16(zlive && a_ok ? zp[jz] : a.zero, lds0 + buf * STAGE + q * 1024);
        zx[jz] += KP;
        zp[jz] += z_step;
        while (zx[jz] >= a.Wo) {
          zx[jz] -= a.Wo;
          zp[jz] += z_row;
          if (++zy[jz] == a.Ho) {
            zy[jz] = 0;
            zp[jz] += z_img;
          }
        }
      }
      lds_dma16(live && b_ok ? ap[j] : a.zero, lds0 + buf * STAGE + TILE_A + q * 1024);
      px[j] += KP;
      ap[j] += a_step;
      while (px[j] >= a.Wo) {
        px[j] -= a.Wo;
        ap[j] += a_row;
        if (++py[j] == a.Ho) {
          py[j] = 0;
          ap[j] += a_img;
        }
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
    const float* tB = tA + KP * BM;
#pragma unroll
    for (int g = 0; g < NJ; ++g) {
      // one ds_read_b64 per operand and pixel pair: lane lrow gets rows / columns 2*lrow and 2*lrow+1 of the
      // wave's 64, so MFMA tile m of an operand covers the interleaved set {2*l + m} (undone in the epilogue);
      // BM = 64: the wave's 32 n columns are read one float per lane
      f32x2 av[4];
      float bv[4][TJ];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int kpix = 2 * (g * 4 + s4) + h;
        av[s4] = *reinterpret_cast<const f32x2*>(tA + kpix * BM + wm * 64 + 2 * lrow);
        if (BM == 128) {
          const f32x2 b2 = *reinterpret_cast<const f32x2*>(tB + kpix * 128 + wn * 64 + 2 * lrow);
          bv[s4][0] = b2[0];
          bv[s4][TJ - 1] = b2[1];
        } else {
          bv[s4][0] = tB[kpix * 128 + wn * 32 + lrow];
        }
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s4][i], bv[s4][j], acc[i][j], 0, 0, 0);
        // a quarter of the next tile's DMA, issued while this group's MFMAs occupy the matrix pipe
        if (s4 == 0 && more) stage(t + 1, (t + 1) & 1, g);
      }
    }
  }

  float* slab = a.slabs + (long long)split * a.Cout * Ntot;
  if (BM == 128) {
    // acc[i][j][r] = D[o = o0 + wm*64 + 2*row(r,h) + i][n = n0 + wn*64 + 2*lrow + j]: the two column tiles of
    // a lane are adjacent in memory -> 8-B stores, 256 B contiguous per half-wave
    const int n = n0 + wn * 64 + 2 * lrow;  // even; Ntot is a multiple of 32
    if (n < Ntot) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + i;
          if (o < a.Cout) {
            f32x2 v = {acc[i][0][r], acc[i][TJ - 1][r]};
            *reinterpret_cast<f32x2*>(slab + (long long)o * Ntot + n) = v;
          }
        }
    }
  } else {
    // acc[i][0][r] = D[o = o0 + 2*row(r,h) + i][n = n0 + wn*32 + lrow]: 128 B contiguous per half-wave
    const int n = n0 + wn * 32 + lrow;
    if (n < Ntot) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = o0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + i;
          if (o < a.Cout) slab[(long long)o * Ntot + n] = acc[i][0][r];
        }
    }
  }
#endif
}

hipError_t vy_launch_wgrad(const WgradArgs& a, hipStream_t s) {
  if (a.Cin % 32 != 0 || a.k_per_split % 32 != 0 || a.splits < 1 || (a.z_cs & 3) || (a.a_cs & 3) || (a.a_co & 3))
    return hipErrorInvalidValue;
  const int Ntot = a.k * a.k * a.Cin;
  const int tiles_n = (Ntot + 127) / 128;
  if (a.Cout <= 64)
    hipLaunchKernelGGL((wgrad_kernel<32, 64>), dim3(tiles_n, a.splits), dim3(256), 0, s, a, tiles_n);
  else
    hipLaunchKernelGGL((wgrad_kernel<32, 128>), dim3((a.Cout + 127) / 128 * tiles_n, a.splits), dim3(256), 0, s, a,
                       tiles_n);
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
