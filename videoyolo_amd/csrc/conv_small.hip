// conv_small.hip — the same implicit-GEMM convolution as conv_igemm.hip (same ConvArgs, same zero-bordered NHWC
// planes, same fused epilogue: folded BatchNorm / bias, LeakyReLU, residual, x2-replicated cropped store, per-tile
// batch statistics) on 16x16 wave tiles: v_mfma_f32_16x16x4_f32.
//
// Why a second tiling.  The reference's default inference call is ONE frame (detect_yolo3.py:55 `batch_size` 1, loop
// :209-222), and at small batches a layer's output is smaller than the chip: a 19x19 x 1024-channel map is 361 wave
// tiles of 32x32 for 1024 SIMDs, and what bounds the launch is then the fma chain of ONE wave per busy SIMD —
// K/2 dependent v_mfma_f32_32x32x2_f32 of 64 cycles each (a K = 4608 layer: 61 us for 22 us of work).  A 16x16 tile
// has a quarter of the outputs per wave, so four times the waves, and its chain is K/4 instructions of 32 cycles
// (issue; 40 dependent): every SIMD gets work and each works through a chain a quarter as long.
//
// Numerics are unchanged — this is what makes the variant admissible.  v_mfma_f32_16x16x4_f32 is an exact fp32 fma
// chain over its four k slots in slot order (tools/probe/mfma16_order_probe.hip: 51 200 / 51 200 bit-equal), and
// lane quarter q = lane / 16 IS slot q.  The pinned order of conv_igemm.hip — per output ONE chain over the taps
// (kh, kw) and inside a tap the channels 0,4,1,5,2,6,3,7 of each aligned group of 8 — is kept by feeding the first
// MFMA of a group the channels (0,4,1,5) and the second (2,6,3,7): quarter q reads the 16-byte chunk that holds
// channels 4*(q&1) .. +3 of the group and takes elements (q>>1) and (q>>1)+2 of it.  Those two selects per fragment
// are vector instructions in the k-loop, which conv_igemm.hip cannot afford; here they are free — the dependent
// 16x16x4 chain leaves 8 of every 40 cycles unused (tools/probe/small_tile_probe.hip: 124.6 TFLOP/s with one wave per
// SIMD with and without them, 132 with two).  Every parity test runs bit-exact through either kernel.
//
// Tiling: 4 waves, block tile 32x32 / 32x64 (wave tile 16x16 / 16x32); per 32-channel sub-step the A and W
// rows (128 B each) arrive by LDS-DMA exactly as in conv_igemm.hip (same XOR swizzle: the ds_read_b128 of a quarter's
// 16 rows at one chunk is conflict-free) into a ring of NS sub-step buffers; one barrier per sub-step; a wave reads the
// fragments of sub-step t+1 into registers while the MFMAs of sub-step t run, with the LDS-DMA of sub-step t+NS
// issued between its MFMA groups.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "../../include/vy_math.h"

#include "conv_device.h"

// tools/probe/conv_small_probe.hip compiles this file with -DVY_S16_ABLATE=<bits> to time the k-loop with parts
// removed (results are then wrong; never defined in the library): 1 no LDS-DMA after the prologue, 2 no MFMAs,
// 4 no fragment reads, 8 no per-stage barrier
#ifndef VY_S16_ABLATE
#define VY_S16_ABLATE 0
#endif

namespace {

// wait until at most `later` sub-steps' worth of this wave's LDS-DMA instructions (D each) are still in flight
// (loads return in order); `later` is wave-uniform, 0 .. 5
template <int D>
__device__ __forceinline__ void wait_vm(int later) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(5 * D <= 63, "6-bit vmcnt");
  switch (later) {
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * D) : "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * D) : "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * D) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#endif
}

template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(256, 2) void conv_s16_kernel(const ConvArgs a, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NW = WM * WN, NT = NW * 64;
  static_assert(NW == 4, "4 waves");
  static_assert(BM % 32 == 0 && BN % 32 == 0, "DMA split: BM/32 A and BN/32 W instructions per wave per sub-step");
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;          // 16x16 accumulators per wave
  static_assert(TM >= 1 && TN >= 1 && TM * 16 * WM == BM && TN * 16 * WN == BN, "tile");
  constexpr int A_PW = BM / 32, B_PW = BN / 32;                // LDS-DMA instructions per wave per sub-step
  constexpr int SUB_A = BM * 128, SUB = (BM + BN) * 128;       // bytes of one 32-channel sub-step: A rows | W rows
  constexpr int D = A_PW + B_PW;                               // DMA instructions per wave per sub-step
  static_assert(NS >= 3 && NS <= 6, "wait_vm: up to NS - 1 later sub-steps in flight");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * SUB + 2 * BM * 8];
  long long* in_off = reinterpret_cast<long long*>(smem + NS * SUB);
  unsigned* o_off = reinterpret_cast<unsigned*>(in_off + BM);  // epilogue row tables: see conv_igemm.hip
  unsigned* r_off = o_off + BM;
  constexpr unsigned kInvalidRow = 0x80000000u;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int q = lane >> 4, l16 = lane & 15;
  const bool hi = (q >> 1) != 0;

  // XCD-aware tile order (conv_igemm.hip): a contiguous run of tiles per XCD, n fastest
  int v;
  {
    const int nblk = gridDim.x, L = blockIdx.x;
    const int qq = nblk >> 3, r = nblk & 7, xcd = L & 7, idx = L >> 3;
    v = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  }
  const int tile_m = v / tiles_n, tile_n = v - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  long long pix0;
  {
    const int t = (int)fd_div((unsigned)m0, a.fd_lw);
    const int x = m0 - t * a.LW;
    const int b = (int)fd_div((unsigned)t, a.fd_lh);
    const int y = t - b * a.LH;
    const long long p = (long long)(b * a.o_Hp + y * a.o_s + a.o_oy) * a.o_Wp + x * a.o_s + a.o_ox;
    pix0 = ((long long)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)p);
  }
  for (int rr = tid; rr < BM; rr += NT) {
    const int m = m0 + rr;
    const int mm = m < a.M ? m : a.M - 1;
    const int t = (int)fd_div((unsigned)mm, a.fd_lw);
    const int x = mm - t * a.LW;
    const int b = (int)fd_div((unsigned)t, a.fd_lh);
    const int y = t - b * a.LH;
    in_off[rr] = ((long long)(b * a.a_Hp + y * a.a_s + a.a_oy) * a.a_Wp + x * a.a_s + a.a_ox) * a.a_cs + a.a_co;
    const unsigned rel = (unsigned)(((long long)(b * a.o_Hp + y * a.o_s + a.o_oy) * a.o_Wp + x * a.o_s + a.o_ox) - pix0);
    unsigned oo_row = (m < a.M) ? rel * (unsigned)a.o_cs * 4u : kInvalidRow;
    if (a.ups == 2 && m < a.M) oo_row |= (2 * x + 1 >= a.o_Wp - 2 ? 1u : 0u) | (2 * y + 1 >= a.o_Hp - 2 ? 2u : 0u);
    o_off[rr] = oo_row;
    r_off[rr] = (m < a.M) ? rel * (unsigned)a.r_cs * 4u : kInvalidRow;
  }
  __syncthreads();

  // LDS-DMA sources: wave-uniform base + per-lane byte offset relative to the tile's first row
  unsigned a_voff[A_PW], b_voff[B_PW];
  const long long in_off0 = in_off[0];
  const float* a_sbase;
  {
    const unsigned long long p = (unsigned long long)(a.in + in_off0);
    a_sbase = (const float*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)p));
  }
  const int wK = a.w_taps * a.w_cin;
#pragma unroll
  for (int j = 0; j < A_PW; ++j) {
    const int row = (j * NW + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[j] = (unsigned)((in_off[row] - in_off0) * 4 + chunk * 16);
  }
#pragma unroll
  for (int j = 0; j < B_PW; ++j) {
    const int row = (j * NW + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int n = n0 + row;
    n = n < a.N ? n : a.N - 1;
    b_voff[j] = (unsigned)((n - n0) * wK * 4 + chunk * 16);
  }
  const float* b_sbase = a.w + (long long)n0 * wK;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const int cchunks = a.Kc >> 5;
  const int T = a.ntaps * cchunks;          // 32-channel sub-steps of the k-loop

  // wave-uniform (tap, channel chunk) state of the next sub-step to be fetched; sub-step t lives in buffer t % NS
  int n_tap = 0, n_cc = 0;
  int a_koff = 0;
  long long b_koff = 0;
  auto next_sub = [&]() {
    const int tdy = (int)((a.pk_dy >> (2 * n_tap)) & 3u) - 1, tdx = (int)((a.pk_dx >> (2 * n_tap)) & 3u) - 1;
    const int tw = (int)((a.pk_w >> (4 * n_tap)) & 15ull);
    a_koff = (tdy * a.a_Wp + tdx) * a.a_cs + n_cc * 32;
    b_koff = tw * a.w_cin + n_cc * 32;
    if (++n_cc == cchunks) {
      n_cc = 0;
      ++n_tap;
    }
  };
  // DMA instruction idx (0 .. D-1) of the sub-step whose offsets next_sub() computed last, into buffer `buf`
  auto dma = [&](int buf, int idx) {
    const unsigned base = lds0 + buf * SUB;
#pragma unroll
    for (int j = 0; j < A_PW; ++j)
      if (idx == j) lds_dma16_s(a_voff[j], a_sbase + a_koff, base + (j * NW + wave) * 1024);
#pragma unroll
    for (int j = 0; j < B_PW; ++j)
      if (idx == A_PW + j) lds_dma16_s(b_voff[j], b_sbase + b_koff, base + SUB_A + (j * NW + wave) * 1024);
  };

  // fragment rows inside a sub-step: byte address row * 128 + (chunk ^ swizzle(row)) * 16, chunk = 2g + (q & 1)
  int a_row[TM], b_row[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_row[i] = (wm * TM + i) * 16 + l16;
#pragma unroll
  for (int j = 0; j < TN; ++j) b_row[j] = (wn * TN + j) * 16 + l16;

  // the raw 16-byte fragments of one sub-step (4 groups of 8 channels), before the per-quarter select
  struct Frag {
    f32x4 a[4][TM], b[4][TN];
  };
  auto load_sub = [&](Frag& f, const unsigned char* sub) {
    if (VY_S16_ABLATE & 4) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = a_row[i];
        f.a[g][i] = *reinterpret_cast<const f32x4*>(sub + row * 128 + (((2 * g + (q & 1)) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = b_row[j];
        f.b[g][j] = *reinterpret_cast<const f32x4*>(sub + SUB_A + row * 128 + (((2 * g + (q & 1)) ^ ((row >> 1) & 7)) << 4));
      }
    }
  };
  // the per-quarter select: ONE v_cndmask_b32.  Written plainly as `hi ? v[1] : v[0]`, hipcc canonicalises it to a
  // variable-index element extract and emits three compares + three selects per value (9.4 M vector instructions per
  // launch against 1.8 M MFMAs on the 19x19 layer: the kernel was VALU-bound).  The empty asm makes the two elements
  // opaque scalars, so the select stays a select — and stays a compiler-generated VALU instruction: an inline-asm
  // v_cndmask is invisible to the hazard recognizer, which then leaves out the wait states between a VALU write and
  // the MFMA that reads it (measured: NaNs).
  auto pick = [&](float lo_half, float hi_half) -> float {
    asm("" : "+v"(lo_half), "+v"(hi_half));
    return hi ? hi_half : lo_half;
  };
  // 4 groups of MFMAs on the fragments of sub-step t.  Behind group 0: the barrier of sub-step t + 1 and the reads of its
  // fragments (`after_group0`); behind groups 1 .. 3 a share each of the LDS-DMA of sub-step t + NS, into the buffer
  // sub-step t occupied (every wave finished reading it before that barrier)
  auto mfma_sub = [&](const Frag& f, int t, auto&& after_group0) {
    const bool fetch = t + NS < T && !(VY_S16_ABLATE & 1);
    if (fetch) next_sub();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float x0[TM], x1[TM], y0[TN], y1[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        x0[i] = pick(f.a[g][i][0], f.a[g][i][1]);
        x1[i] = pick(f.a[g][i][2], f.a[g][i][3]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        y0[j] = pick(f.b[g][j][0], f.b[g][j][1]);
        y1[j] = pick(f.b[g][j][2], f.b[g][j][3]);
      }
      if (VY_S16_ABLATE & 2) {  // keep the reads and selects alive
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(x0[i]), "v"(x1[i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(y0[j]), "v"(y1[j]));
      } else {
        // chain order per output: slots (q = 0..3) of the first MFMA = channels 8g + (0,4,1,5), second 8g + (2,6,3,7)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i], y0[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i], y1[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (g == 0) {
        // (here, not at the top of the sub-step: hipcc cannot carry the LDS counter across the loop edge and waits
        // for lgkmcnt(0) before the first use of this sub-step's fragments — which must not include the next one's reads)
        after_group0();
      } else if (fetch) {
        constexpr int lo[3] = {0, (D + 2) / 3, (2 * D + 2) / 3}, hi_[3] = {(D + 2) / 3, (2 * D + 2) / 3, D};
#pragma unroll
        for (int idx = 0; idx < D; ++idx)
          if (idx >= lo[g - 1] && idx < hi_[g - 1]) dma(t % NS, idx);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // before reading sub-step t: it has landed for every wave, and every wave's reads of sub-step t-1 are complete
  // (lds_barrier waits for lgkmcnt(0)) — so mfma_sub(t - 1) may overwrite that buffer with sub-step t - 1 + NS.
  // Issued so far: sub-steps up to t - 2 + NS; those after t may stay in flight.
  auto enter = [&](int t) {
    const int issued_last = t - 2 + NS < T - 1 ? t - 2 + NS : T - 1;
    wait_vm<D>(issued_last - t);
    if (!(VY_S16_ABLATE & 8)) lds_barrier();
  };

  for (int t = 0; t < NS && t < T; ++t) {  // all NS buffers are filled before any matrix work
    next_sub();
#pragma unroll
    for (int idx = 0; idx < D; ++idx) dma(t, idx);
  }
  // Fragment reads run one sub-step ahead of the MFMAs, in two register sets with static parity
  Frag f0 = {}, f1 = {};
  wait_vm<D>(NS - 1 < T - 1 ? NS - 1 : T - 1);  // enter(0): sub-steps 1 .. min(NS, T) - 1 may stay in flight
  lds_barrier();
  load_sub(f0, smem);
  for (int t = 0; t < T; t += 2) {
    mfma_sub(f0, t, [&]() {
      if (t + 1 < T) {
        enter(t + 1);
        load_sub(f1, smem + ((t + 1) % NS) * SUB);
      }
    });
    if (t + 1 >= T) break;
    mfma_sub(f1, t + 1, [&]() {
      if (t + 2 < T) {
        enter(t + 2);
        load_sub(f0, smem + ((t + 2) % NS) * SUB);
      }
    });
  }

  // ---- epilogue (conv_igemm.hip's, on the 16x16 accumulator layout: lane = column l16, rows 4q .. 4q+3)
  constexpr int kRsrcFlags = 0x00020000;
  const __amdgpu_buffer_rsrc_t out_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(a.out + (pix0 * a.o_cs + a.o_co + n0), 0, 0x7fffffff, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res ? a.res + (pix0 * a.r_cs + a.r_co + n0) : a.in), 0, 0x7fffffff, kRsrcFlags);
  const int ups_dx = a.o_cs * 4, ups_dy = a.o_Wp * a.o_cs * 4;
  auto epilogue = [&](auto has_scale_, auto has_shift_, auto leaky_, auto has_res_, auto ups2_) {
    constexpr bool has_scale = decltype(has_scale_)::value, has_shift = decltype(has_shift_)::value;
    constexpr bool leaky = decltype(leaky_)::value, has_res = decltype(has_res_)::value, ups2 = decltype(ups2_)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ncol = (wn * TN + j) * 16 + l16;
      const int n = n0 + ncol;
      const bool nvalid = n < a.N;
      const int nc = nvalid ? n : a.N - 1;
      const unsigned colc = (unsigned)ncol * 4u | (nvalid ? 0u : kInvalidRow);
      float sc = 1.0f, sh = 0.0f;
      if (has_scale) sc = a.scale[nc];
      if (has_scale || has_shift) sh = a.shift[nc];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        unsigned oo[4];
        float rv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wm * TM + i) * 16 + 4 * q + r;
          oo[r] = __builtin_elementwise_add_sat(o_off[row], colc);
          if (has_res) rv[r] = buf_load_f32(res_rsrc, __builtin_elementwise_add_sat(r_off[row], colc));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float vv = acc[i][j][r];
          if (has_scale)
            vv = fmaf(vv, sc, sh);
          else if (has_shift)
            vv = vv + sh;
          if (leaky) vv = vy_leaky(vv);
          if (has_res) vv = vv + rv[r];
          if (!ups2) {
            buf_store_f32(vv, out_rsrc, oo[r], 0);
          } else {
            const unsigned base = oo[r] & ~3u;
            const unsigned no_dx = (oo[r] & 1u) << 31, no_dy = (oo[r] & 2u) << 30;
            buf_store_f32(vv, out_rsrc, base, 0);
            buf_store_f32(vv, out_rsrc, base | no_dx, ups_dx);
            buf_store_f32(vv, out_rsrc, base | no_dy, ups_dy);
            buf_store_f32(vv, out_rsrc, base | no_dx | no_dy, ups_dy + ups_dx);
          }
        }
      }
    }
  };
  {
    using T_ = std::true_type;
    using F_ = std::false_type;
    const bool f_scale = a.scale != nullptr, f_shift = a.shift != nullptr, f_leaky = a.leaky != 0;
    const bool f_res = a.res != nullptr, f_ups2 = a.ups == 2;
    if (f_scale && f_shift && f_leaky && !f_ups2) {
      if (f_res) epilogue(T_{}, T_{}, T_{}, T_{}, F_{});
      else epilogue(T_{}, T_{}, T_{}, F_{}, F_{});
    } else if (f_scale && f_shift && f_leaky && !f_res) {
      epilogue(T_{}, T_{}, T_{}, F_{}, T_{});
    } else if (!f_scale && !f_leaky && !f_ups2) {
      if (f_shift) {
        if (f_res) epilogue(F_{}, T_{}, F_{}, T_{}, F_{});
        else epilogue(F_{}, T_{}, F_{}, F_{}, F_{});
      } else {
        if (f_res) epilogue(F_{}, F_{}, F_{}, T_{}, F_{});
        else epilogue(F_{}, F_{}, F_{}, F_{}, F_{});
      }
    } else {
      __builtin_trap();  // vy_launch_conv_igemm rejects every other combination
    }
  }

  // train-mode BatchNorm: per-tile column sums of the raw accumulators in double, fixed order
  if (a.stats) {
    double s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] = 0.0;
      s2[j] = 0.0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wm * TM + i) * 16 + 4 * q + r;
          const double vv = o_off[row] != kInvalidRow ? (double)acc[i][j][r] : 0.0;
          s1[j] += vv;
          s2[j] += vv * vv;
        }
      // the four quarters hold rows 4q .. 4q+3 of the same column: combine in a fixed order
      s1[j] += __shfl_xor(s1[j], 16);
      s2[j] += __shfl_xor(s2[j], 16);
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
    }
    __syncthreads();  // every wave is past its last LDS tile read
    double* red = reinterpret_cast<double*>(smem);  // [WM][2][BN]
    static_assert(WM * 2 * BN * 8 <= NS * SUB, "statistics scratch fits the tile buffers");
    if (q == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 16 + l16;
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

template <int BM, int BN, int WM, int WN, int NS>
hipError_t launch_s16(const ConvArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL((conv_s16_kernel<BM, BN, WM, WN, NS>), dim3(tiles_m * tiles_n), dim3(256), 0, s, a, tiles_n);
  return hipGetLastError();
}

}  // namespace

// forward launches only (dgrad reads W as the [k][n] operand: conv_igemm.hip); `a` already carries the launcher-filled
// fields (fast divisors, packed tap tables)
hipError_t vy_launch_conv_s16(const ConvArgs& a, int bm, int bn, hipStream_t s) {
  if (a.dgrad) return hipErrorInvalidValue;
  if (bm == 32 && bn == 32) return launch_s16<32, 32, 2, 2, 6>(a, s);
  if (bm == 32 && bn == 64) return launch_s16<32, 64, 2, 2, 5>(a, s);
  // mid-size tiles: 16-row / 16-column granularity to land a small launch on one round of the CUs
  if (bm == 64 && bn == 64) return launch_s16<64, 64, 2, 2, 4>(a, s);
  if (bm == 96 && bn == 64) return launch_s16<96, 64, 2, 2, 4>(a, s);
  if (bm == 64 && bn == 96) return launch_s16<64, 96, 2, 2, 4>(a, s);
  if (bm == 96 && bn == 96) return launch_s16<96, 96, 2, 2, 3>(a, s);
  if (bm == 128 && bn == 64) return launch_s16<128, 64, 2, 2, 3>(a, s);
  return hipErrorInvalidValue;
}
