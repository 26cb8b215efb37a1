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
#include <type_traits>

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

// the same through a wave-uniform base (SGPR pair) + a per-lane 32-bit byte offset (see conv_igemm.hip)
__device__ __forceinline__ void lds_dma16_s(unsigned voff, const float* sbase, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr)
               : "memory", "m0");
#endif
}

// Pixel table of one conv (built once per plan by wgrad_table_kernel): for output pixel p the byte offset of its dz
// vector inside the dz plane and of its (stride-scaled) centre pixel inside the input plane, shifted by one row and
// one column so that every tap offset is non-negative.  Entries M .. pad are "dead" pixels: a zero border pixel of
// the dz plane (contributes 0 to every sum) and a valid input pixel.  With it the k-loop needs no coordinates at all:
// the fp32 MFMA shares the vector pipe (tools/probe/coissue_probe.hip), and the incremental per-lane coordinates of
// round 1 cost 92 vector instructions per 64 MFMAs.
// (round 4) Entries are 32-bit byte offsets RELATIVE to the first pixel of the entry's split (`k_per_split` pixels of the
// reduction go to one block; offsets grow with the pixel index, so they are >= 0 and a split spans far less than
// 4 GiB): the planes themselves may be of any size.  The kernel adds the split's base — wgrad_split_base, the same
// formulas — to its wave-uniform DMA base once per block.  Dead entries point at the bottom-left border pixel of the LAST
// image (zero, and behind every valid pixel) and at the last valid input pixel.
struct WgradPix {
  unsigned long long zo, ao;
};
__device__ __forceinline__ WgradPix wgrad_pixel_offsets(int p, int Ho, int Wo, int z_cs, int a_Hp, int a_Wp, int a_cs, int stride) {
  const int x = p % Wo, t = p / Wo, y = t % Ho, b = t / Ho;
  WgradPix o;
  o.zo = ((unsigned long long)(b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * z_cs * 4ull;
  o.ao = ((unsigned long long)(b * a_Hp + y * stride + 1) * a_Wp + x * stride + 1) * a_cs * 4ull;
  return o;
}

__global__ __launch_bounds__(256) void wgrad_table_kernel(uint2* __restrict__ tab, int M, int n_entries, int Ho, int Wo,
                                                          int z_cs, int a_Hp, int a_Wp, int a_cs, int stride, int B,
                                                          int k_per_split) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n_entries) return;
  const int pp = p < M ? p : M - 1;
  const int split = pp / k_per_split;
  const WgradPix base = wgrad_pixel_offsets(split * k_per_split, Ho, Wo, z_cs, a_Hp, a_Wp, a_cs, stride);
  const WgradPix o = wgrad_pixel_offsets(pp, Ho, Wo, z_cs, a_Hp, a_Wp, a_cs, stride);
  const unsigned long long dead = ((unsigned long long)((B - 1) * (Ho + 2) + Ho + 1) * (Wo + 2)) * z_cs * 4ull;
  uint2 e;
  e.x = (unsigned)((p < M ? o.zo : dead) - base.zo);
  e.y = (unsigned)(o.ao - base.ao);
  tab[p] = e;
}

hipError_t vy_launch_wgrad_table(void* tab, int M, int n_entries, int Ho, int Wo, int z_cs, int a_Hp, int a_Wp, int a_cs,
                                 int stride, int B, int k_per_split, hipStream_t s) {
  if (k_per_split < 1 || B < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(wgrad_table_kernel, dim3((n_entries + 255) / 256), dim3(256), 0, s, static_cast<uint2*>(tab), M,
                     n_entries, Ho, Wo, z_cs, a_Hp, a_Wp, a_cs, stride, B, k_per_split);
  return hipGetLastError();
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
  // Block -> (split, tile).  All tiles of a split read the same pixel range: every n-tile its dz rows, every o-tile the
  // input taps.  Blocks are dealt round-robin over the 8 XCDs (each with its own L2), so in launch order every XCD saw
  // every split and each re-read went out to the fabric (round 2: 22.6 GB per step against 5.2 GB algorithmic).  With
  // a.xcd_order the linear block index is mapped so that an XCD works through a CONTIGUOUS run of (split, tile) pairs,
  // tile fastest — whole splits per XCD, their re-reads served by that XCD's L2.
  int tile_id = blockIdx.x, split = blockIdx.y;
  if (a.xcd_order) {
    const int gx = gridDim.x, nblk = gx * gridDim.y, L = blockIdx.y * gx + blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, idx = L >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    split = v / gx;
    tile_id = v - split * gx;
  }
  const int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;
  const int o0 = tile_m * BM, n0 = tile_n * BN;
  const int Ntot = a.k * a.k * a.Cin;
  const int p_begin = split * a.k_per_split;
  int p_end = p_begin + a.k_per_split;
  if (p_end > a.M) p_end = a.M;
  const int T = (p_end - p_begin + KP - 1) / KP;

  // this lane's fixed column chunk in both tiles
  const int a_chunk = lane % A_CPR, a_sub = lane / A_CPR;  // dz: chunk inside the row, pixel row inside the instruction
  const int ao = o0 + a_chunk * 4;        // dz channel of this chunk
  const int chunk = lane & 31;            // activation tile: 16-B chunk inside a 512-B row
  const int bn = n0 + chunk * 4;          // n column of this chunk
  const bool b_ok = bn < Ntot;            // columns past k*k*Cin: tap 0 / channel 0 (the column is never stored)
  const int tap = b_ok ? bn / a.Cin : 0;
  const int cin = b_ok ? bn - tap * a.Cin : 0;
  const int pad = a.k >> 1;
  const int dy = a.k == 3 ? tap / 3 - pad : 0, dx = a.k == 3 ? tap % 3 - pad : 0;

  f32x16 acc[2][TJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // Staging.  Each lane stages the same pixel rows of both tiles every k-step (activations: row 2*(j*4+wave) + h;
  // dz: row A_RPI*(j*4+wave) + a_sub).  Their byte offsets come from the conv's pixel table, one 8-byte entry per
  // row and k-step, loaded a k-step ahead; the DMA address is a wave-uniform base + (table offset + the lane's
  // constant: its channel chunk / its tap and input channel): one vector add per DMA instruction and nothing else.
  // Rows past the end of the pixel range are dead table entries (a zero dz pixel: they add 0).  Lanes whose chunk lies
  // past Cout / past k*k*Cin read whatever follows in memory: those rows / columns of the tile are never stored,
  // and an MFMA output element depends on its own row and column only.
  const uint2* tab = a.tab + p_begin;                                      // wave-uniform
  // the split's first pixel: the table's offsets are relative to it (wave-uniform; 32-bit divisions + readfirstlane,
  // sk-kernel lesson: a 64-bit division is expanded into vector code whose results stop counting as uniform)
  const WgradPix sb = wgrad_pixel_offsets(p_begin, a.Ho, a.Wo, a.z_cs, a.a_Hp, a.a_Wp, a.a_cs, a.stride);
  auto uni64 = [](unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)v);
  };
  const float* z_base = a.dz + (uni64(sb.zo) >> 2);
  const float* a_base = a.a + (uni64(sb.ao) >> 2) + a.a_co - (long long)(a.a_Wp + 1) * a.a_cs;  // tap offsets become >= 0
  const unsigned zc = (unsigned)ao * 4u;
  const unsigned ac = (unsigned)((((dy + 1) * a.a_Wp + (dx + 1)) * a.a_cs + cin) * 4);
  const int zrow0 = A_RPI * wave + a_sub, arow0 = 2 * wave + h;  // + 4 * A_RPI * j / + 8 * j
  unsigned tz[NJA], ta[NJ];
  auto load_offsets = [&](int t) {
#pragma unroll
    for (int j = 0; j < NJA; ++j) tz[j] = tab[t * KP + zrow0 + 4 * A_RPI * j].x;
#pragma unroll
    for (int j = 0; j < NJ; ++j) ta[j] = tab[t * KP + arow0 + 8 * j].y;
  };
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  // DMA of tile t + 1 (whose offsets are in tz / ta) into buffer buf: the activation instruction j of this wave and
  // (j < NJA) the dz instruction j
  auto stage = [&](int buf, int j) {
    const int q = j * 4 + wave;
    if (j < NJA) lds_dma16_s(tz[j < NJA ? j : 0] + zc, z_base, lds0 + buf * STAGE + q * 1024);
    lds_dma16_s(ta[j] + ac, a_base, lds0 + buf * STAGE + TILE_A + q * 1024);
  };

  if (T > 0) {
    load_offsets(0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) stage(0, j);
    if (T > 1) load_offsets(1);
  }
  // One k-step on the tile in buffer PAR (a compile-time constant: the loop below is unrolled by two, so the buffer
  // offset folds into the ds_read immediates — round 2's loop added it to every fragment address, 38 vector adds per
  // 32 MFMAs in the 64-row instance).  The fragments of group g + 1 are read while the MFMAs of group g run (two
  // register sets): before, every group read its fragments and waited for them, four exposed LDS latencies per
  // k-step (SQ counters of round 3: 55 % MFMA-busy for the 64-row instance, 76 % for the 128-row one).
  struct Frag {
    f32x2 av[4];
    float bv[4][TJ];
  };
  auto kstep = [&](auto par_, const int t) {
    constexpr int PAR = decltype(par_)::value;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of tile t has landed
    __syncthreads();
    const bool more = t + 1 < T;
    const float* tA = reinterpret_cast<const float*>(smem + PAR * STAGE);
    const float* tB = tA + KP * BM;
    // one ds_read_b64 per operand and pixel pair: lane lrow gets rows / columns 2*lrow and 2*lrow+1 of the
    // wave's 64, so MFMA tile m of an operand covers the interleaved set {2*l + m} (undone in the epilogue);
    // BM = 64: the wave's 32 n columns are read one float per lane
    auto load_group = [&](Frag& f, const int g) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const int kpix = 2 * (g * 4 + s4) + h;
        f.av[s4] = *reinterpret_cast<const f32x2*>(tA + kpix * BM + wm * 64 + 2 * lrow);
        if (BM == 128) {
          const f32x2 b2 = *reinterpret_cast<const f32x2*>(tB + kpix * 128 + wn * 64 + 2 * lrow);
          f.bv[s4][0] = b2[0];
          f.bv[s4][TJ - 1] = b2[1];
        } else {
          f.bv[s4][0] = tB[kpix * 128 + wn * 32 + lrow];
        }
      }
    };
    Frag f0, f1;
    load_group(f0, 0);
#pragma unroll
    for (int g = 0; g < NJ; ++g) {
      Frag& cur = (g & 1) ? f1 : f0;
      Frag& nxt = (g & 1) ? f0 : f1;
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < NJ) load_group(nxt, g + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.av[s4][i], cur.bv[s4][j], acc[i][j], 0, 0, 0);
        // a quarter of the next tile's DMA, issued while this group's MFMAs occupy the matrix pipe
        if (s4 == 0 && more) stage(PAR ^ 1, g);
      }
    }
    if (t + 2 < T) load_offsets(t + 2);  // consumed by the stage() calls of the next k-step, after its vmcnt(0)
  };
  {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 1 < T; t += 2) {
      kstep(P0{}, t);
      kstep(P1{}, t + 1);
    }
    if (t < T) kstep(P0{}, t);
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

// 64-row tiles for Cout <= 64 (no zero padding in the MFMA rows) and for weight tensors of so few 128-row tiles that
// the planner would cut the pixel range into dozens of splits of 64 KB slabs: every 1x1 layer and tensors of at most
// VY_WGRAD_BM64_TILES (8) 128-row tiles (measured in the step: +0.4 %; larger 3x3 tensors lose)
int vy_wgrad_tile_rows(int Cout, int k, int Cin) {
  static const int max_tiles = getenv("VY_WGRAD_BM64_TILES") ? atoi(getenv("VY_WGRAD_BM64_TILES")) : 8;
  if (Cout <= 64) return 64;
  if (max_tiles < 0) return 128;  // experiment switch: the round-1 rule
  const int tiles128 = ((Cout + 127) / 128) * ((k * k * Cin + 127) / 128);
  return (k == 1 || tiles128 <= max_tiles) ? 64 : 128;
}

hipError_t vy_launch_wgrad(const WgradArgs& a_in, hipStream_t s) {
  WgradArgs a = a_in;
  static const int xcd_order = getenv("VY_WGRAD_XCD") ? atoi(getenv("VY_WGRAD_XCD")) : 1;
  a.xcd_order = xcd_order;
  if (a.Cin % 32 != 0 || a.k_per_split % 32 != 0 || a.splits < 1 || (a.z_cs & 3) || (a.a_cs & 3) || (a.a_co & 3) || !a.tab)
    return hipErrorInvalidValue;
  const int Ntot = a.k * a.k * a.Cin;
  const int tiles_n = (Ntot + 127) / 128;
  if (vy_wgrad_tile_rows(a.Cout, a.k, a.Cin) == 64)
    hipLaunchKernelGGL((wgrad_kernel<32, 64>), dim3((a.Cout + 63) / 64 * tiles_n, a.splits), dim3(256), 0, s, a, tiles_n);
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
