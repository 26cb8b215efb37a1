// conv_ring.hip — forward implicit-GEMM convolution, LDS-DMA ring variant of conv_igemm.hip.
//
// Same GEMM view, plane layout, tap table, epilogue and — to the bit — the same summation order as
// conv_igemm_kernel (taps ascending; channels 0,4,1,5,2,6,3,7 per aligned group of 8).  What changes is
// the staging: instead of a double buffer of 32-channel tiles (prefetch distance one tile, whose last
// DMA's latency shows at every tile-end wait), the 64 KiB hold a RING OF FOUR 16-channel stages
// (64 B per row), refilled three stages ahead and retired with a COUNTED s_waitcnt vmcnt(n), never 0
// inside the loop.  tools/probe/dma_probe.hip: with every k-step reading fresh HBM lines the double
// buffer sustains 134 TF, the ring 142 TF.
//
// Per stage s (2 groups of 8 channels = 8 MFMA steps per 32x32 tile):
//   group 0: MFMA steps 0,1 | vmcnt(one stage) + barrier: stage s+1 visible to all waves and every wave is
//            done with stage s-1 | DMA of stage s+3 into the slot of stage s-1 | step 2 | ds_reads of
//            group 1 | step 3
//   group 1: MFMA steps 0,1,2 | ds_reads of stage s+1's group 0 | step 3
// so neither the LDS latency nor the barrier sits between two stages.
// 16-B chunk c of row r is stored at chunk position c ^ ((r>>2)&3) (source-side swizzle; rows are 64 B, four
// rows per 256-B bank row): ds_read_b128 of 32 rows is bank-conflict free.
#include <cstdlib>

#include "kernels.h"
#include "../../include/vy_math.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void ring_dma16(const float* gptr, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory", "m0");
#endif
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_ring_kernel(const ConvArgs a, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;  // one 16-channel stage
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int A_INSTR = BM / 16 / NW, B_INSTR = BN / 16 / NW;  // DMA instructions per wave per stage
  constexpr int PER = A_INSTR + B_INSTR;
  constexpr bool DGRAD = false;
  static_assert(NW == 4, "4 waves");
  static_assert(A_INSTR >= 1 && B_INSTR >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * STAGE + 2 * BM * 8];
  long long* in_off = reinterpret_cast<long long*>(smem + 4 * STAGE);
  long long* o_pix = in_off + BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5;
  const int lrow = lane & 31;

  int v;
  {
    const int nblk = gridDim.x, L = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, idx = L >> 3;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = v / tiles_n, tile_n = v - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  for (int rr = tid; rr < BM; rr += NT) {
    const int m = m0 + rr;
    const int mm = m < a.M ? m : a.M - 1;
    const int x = mm % a.LW;
    const int t = mm / a.LW;
    const int y = t % a.LH;
    const int b = t / a.LH;
    in_off[rr] = ((long long)(b * a.a_Hp + y * a.a_s + a.a_oy) * a.a_Wp + x * a.a_s + a.a_ox) * a.a_cs + a.a_co;
    o_pix[rr] = (m < a.M) ? ((long long)(b * a.o_Hp + y * a.o_s + a.o_oy) * a.o_Wp + x * a.o_s + a.o_ox) : -1;
  }
  __syncthreads();

  // per-lane DMA source pointers: one instruction = 16 rows x 64 B; lane -> row (lane>>2), chunk slot lane&3
  const float* a_src[A_INSTR];
  const float* b_src[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (j * NW + wave) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    a_src[j] = a.in + in_off[row] + chunk * 4;
  }
  const int wK = a.w_taps * a.w_cin;
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int row = (j * NW + wave) * 16 + (lane >> 2);
    const int chunk = (lane & 3) ^ ((row >> 2) & 3);
    int n = n0 + row;
    n = n < a.N ? n : a.N - 1;
    b_src[j] = a.w + (long long)n * wK + chunk * 4;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const int c16 = a.Kc >> 4;        // 16-channel stages per tap
  const int NS = a.ntaps * c16;     // stages in all

  // wave-uniform description of the stage the NEXT dma() call loads (advanced incrementally)
  int d_tap = 0, d_c = 0, d_st = 0;
  auto dma = [&]() {
    const int tdy = (int)((a.pk_dy >> (2 * d_tap)) & 3u) - 1, tdx = (int)((a.pk_dx >> (2 * d_tap)) & 3u) - 1;
    const int tw = (int)((a.pk_w >> (4 * d_tap)) & 15ull);
    const long long a_koff = (long long)(tdy * a.a_Wp + tdx) * a.a_cs + d_c * 16;
    const int b_koff = tw * a.w_cin + d_c * 16;
    const unsigned slot = lds0 + (d_st & 3) * STAGE;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) ring_dma16(a_src[j] + a_koff, slot + (j * NW + wave) * 1024);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) ring_dma16(b_src[j] + b_koff, slot + A_BYTES + (j * NW + wave) * 1024);
    ++d_st;
    // past the last stage the same (valid) source is loaded again into a slot nobody reads any more: the
    // DMA count per stage stays constant, which is what the counted waits below rely on
    if (d_st < NS) {
      if (++d_c == c16) {
        d_c = 0;
        ++d_tap;
      }
    }
  };

  f32x4 af[2][TM], bf[2][TN];
  auto load_group = [&](int st, int g, int buf) {
    const unsigned char* sA = smem + (st & 3) * STAGE;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = (wm * TM + i) * 32 + lrow;
      af[buf][i] = *reinterpret_cast<const f32x4*>(sA + row * 64 + (((2 * g + h) ^ ((row >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = (wn * TN + j) * 32 + lrow;
      bf[buf][j] = *reinterpret_cast<const f32x4*>(sB + row * 64 + (((2 * g + h) ^ ((row >> 2) & 3)) << 4));
    }
  };
  auto mfma_step = [&](int buf, int st) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][i][st], bf[buf][j][st], acc[i][j], 0, 0, 0);
  };

  // prologue: three stages in flight, stage 0 landed and visible, its group 0 in registers
  dma();
  dma();
  dma();
  wait_vmcnt<2 * PER>();
  __syncthreads();
  load_group(0, 0, 0);

  for (int st = 0; st < NS; ++st) {
    // ---- group 0 of stage st (fragments in buffer 0)
    mfma_step(0, 0);
    mfma_step(0, 1);
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<PER>();   // all but the youngest stage (st+2) have landed: stage st+1 is complete
    __syncthreads();     // ... for every wave; and every wave has finished reading stage st-1
    dma();               // stage st+3 -> slot of stage st-1
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(0, 2);
    __builtin_amdgcn_sched_barrier(0);
    load_group(st, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(0, 3);
    // ---- group 1 (buffer 1)
    mfma_step(1, 0);
    mfma_step(1, 1);
    mfma_step(1, 2);
    __builtin_amdgcn_sched_barrier(0);
    load_group(st + 1, 0, 0);  // past the end: reads a slot that still holds valid (stale) data, unused
    __builtin_amdgcn_sched_barrier(0);
    mfma_step(1, 3);
  }
  wait_vmcnt<0>();  // the trailing (redundant) DMAs must be out of LDS before smem is reused below
  __syncthreads();

  // epilogue: affine (folded BN or bias) -> leaky -> + addend -> store (x1 or x2-replicated).
  // Loads of the addend are unconditional (invalid rows / columns read a clamped, valid address) so
  // that all 16*TM of a column tile are in flight together; only the stores are predicated.
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + lrow;
    const bool nvalid = n < a.N;
    const int nc = nvalid ? n : a.N - 1;
    float sc = 1.0f, sh = 0.0f;
    if (a.scale) sc = a.scale[nc];
    if (a.shift) sh = a.shift[nc];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      long long op[16];
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        op[r] = o_pix[row];
      }
      if (a.res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = a.res[(op[r] < 0 ? 0 : op[r]) * a.r_cs + a.r_co + nc];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float vv = acc[i][j][r];
        if (a.scale)
          vv = fmaf(vv, sc, sh);
        else if (a.shift)
          vv = vv + sh;
        if (a.leaky) vv = vy_leaky(vv);
        if (a.res) vv = vv + rv[r];
        if (op[r] >= 0 && nvalid) {
          float* o = a.out + op[r] * a.o_cs + a.o_co + n;
          o[0] = vv;
          if (a.ups == 2) {
            o[a.o_cs] = vv;
            o[(long long)a.o_Wp * a.o_cs] = vv;
            o[(long long)(a.o_Wp + 1) * a.o_cs] = vv;
          }
        }
      }
    }
  }

  // train-mode BatchNorm: per-tile column sums of the raw accumulators (deterministic: fixed
  // order inside the tile, tiles are combined in order by the finalize kernel)
  if (a.stats) {
    // double accumulation: var = E[x^2] - mean^2 cancels badly in fp32 when |mean| >> std, and the
    // batch statistics must round to the same fp32 mean / var as the CPU checker's
    double s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] = 0.0;
      s2[j] = 0.0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const double vv = o_pix[row] >= 0 ? (double)acc[i][j][r] : 0.0;
          s1[j] += vv;
          s2[j] += vv * vv;
        }
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
    }
    __syncthreads();  // every wave is past its last LDS tile read
    double* red = reinterpret_cast<double*>(smem);  // [WM][2][BN]
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + lrow;
        red[(wm * 2 + 0) * BN + col] = s1[j];
        red[(wm * 2 + 1) * BN + col] = s2[j];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.N) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        t1 += red[(w * 2 + 0) * BN + tid];
        t2 += red[(w * 2 + 1) * BN + tid];
      }
      a.stats[((long long)tile_m * 2 + 0) * a.N + n0 + tid] = t1;
      a.stats[((long long)tile_m * 2 + 1) * a.N + n0 + tid] = t2;
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM, int BN, int WM, int WN>
static hipError_t launch_ring(const ConvArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((conv_ring_kernel<BM, BN, WM, WN>), dim3(tiles_m * tiles_n), dim3(WM * WN * 64), 0, s, a, tiles_n);
  return hipGetLastError();
}

// forward launches with a 128x128 / 128x64 / 64x64 tile (a must carry the packed tap tables)
hipError_t vy_launch_conv_ring(const ConvArgs& a, int bm, int bn, hipStream_t s) {
  if (a.dgrad || a.Kc % 16 != 0) return hipErrorInvalidValue;
  if (bm == 128 && bn == 128) return launch_ring<128, 128, 2, 2>(a, s);
  if (bm == 128 && bn == 64) return launch_ring<128, 64, 2, 2>(a, s);
  if (bm == 64 && bn == 64) return launch_ring<64, 64, 2, 2>(a, s);
  return hipErrorInvalidValue;
}
