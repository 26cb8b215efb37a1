// PROTOTYPE (not product code): the split-fp32 3x3 stride-1 conv as a 1-D Winograd F(2, 3) along x — 1.5x fewer
// multiplications.  One block = 128 pixel PAIRS (x0 = 2 xp, x0 + 1) x 64 output channels; four phases
// xi = 0..3, each the split kernel's pipelined k-loop over (dy, 16-channel chunk) with
//     V_0 = d0 - d2,  V_1 = d1 + d2,  V_2 = d2 - d1,  V_3 = d1 - d3        (d_j = the pixel at x0 - 1 + j of row y + dy)
// formed in registers from two pixel loads, split into the three bf16 planes and multiplied with the pre-transformed,
// pre-split weights U_0 = g0, U_1 = (g0 + g1 + g2) / 2, U_2 = (g0 - g1 + g2) / 2, U_3 = g2 into its own accumulator set M_xi;
// epilogue  Y(x0) = M_0 + M_1 + M_2,  Y(x0 + 1) = M_1 - M_2 - M_3,  then affine + leaky.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-inline-asm \
//       -I include -o conv_wino_probe tools/probe/conv_wino_probe.hip videoyolo_amd/csrc/conv_small.hip
//   ./conv_wino_probe B H Cin Cout [reps=20]
#include "../../videoyolo_amd/csrc/conv_split.hip"
#include "../../videoyolo_amd/csrc/conv_igemm.hip"

#include <cmath>
#include <cstring>
#include <vector>

#define CK(x)                                                 \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                               \
    }                                                         \
  } while (0)

struct WinoArgs {
  const float* in;            // [B][H+2][W+2][Cin], zero border
  float* out;                 // [B][H+2][W+2][Cout]
  const unsigned char* wimg;  // 4 images (xi) of vy_split_weight_bytes(Cout, 3, Cin) bytes each
  long long wimg_bytes;
  const float *scale, *shift;
  int B, H, W, Cin, Cout, Wp2, Mp;  // Wp2 = ceil(W / 2), Mp = B * H * Wp2 pairs
};

#ifndef WINO_BM
#define WINO_BM 128  // pairs per block; the block is WINO_BM pairs x (8192 / WINO_BM) channels: 128 x 64 or 64 x 128
#endif
#define WINO_BN (8192 / WINO_BM)

__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const WinoArgs a, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = WINO_BM, BN = WINO_BN, NSA = 3, NSW = 3, WM = 2, WN = 2, NW = 4, NT = 256;
  constexpr bool A_PART = BM * 2 < NT;  // 64 pairs: waves 0-1 stage A
  constexpr int A_PL = BM * 32, W_PL = BN * 32, A_ST = 3 * A_PL, W_ST = 3 * W_PL, W_BASE = NSA * A_ST;
  constexpr int W_TOTAL = (BN / 32) * 3, W_INSTR = (W_TOTAL + NW - 1) / NW;
  constexpr int TMs = BM / WM / 32, TNs = BN / WN / 32;  // 2 x 1
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSA * A_ST + NSW * W_ST + BM * 16];
  long long* in_off = reinterpret_cast<long long*>(smem + NSA * A_ST + NSW * W_ST);
  long long* o_off = in_off + BM;  // element offset of output pixel x0 (channel 0); < 0: invalid row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, lrow = lane & 31;
  const int cch = a.Cin >> 4;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Wp = a.W + 2;
  for (int rr = tid; rr < BM; rr += NT) {
    const int m = m0 + rr, mm = m < a.Mp ? m : a.Mp - 1;
    const int xp = mm % a.Wp2, t = mm / a.Wp2, y = t % a.H, b = t / a.H;
    const long long pix = ((long long)(b * (a.H + 2) + y + 1) * Wp + 2 * xp + 1);  // centre of x0
    in_off[rr] = pix * a.Cin;
    o_off[rr] = m < a.Mp ? pix * a.Cout : -1;
  }
  __syncthreads();
  const bool a_active = !A_PART || wave < BM * 2 / 64;
  const int row_s = (tid & (BM * 2 - 1)) >> 1, oct_s = tid & 1;
  const float* a_ptr = a.in + in_off[row_s] + oct_s * 8;
  const unsigned a_lds = (unsigned)(row_s * 32 + (VY_SPLIT_SLOT(row_s, oct_s) << 4));
  const int KS = 3 * cch;
  unsigned w_voff[W_INSTR], w_lds[W_INSTR];
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j) {
    const int q = j * NW + wave, g = q / 3, p = q - g * 3;
    w_voff[j] = (unsigned)(g * KS * 3072 + p * 1024 + lane * 16);
    w_lds[j] = (unsigned)(W_BASE + p * W_PL + g * 1024);
  }
  f32x16 acc[4][TMs][TNs];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int i = 0; i < TMs; ++i)
#pragma unroll
      for (int j = 0; j < TNs; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[x][i][j][r] = 0.0f;
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const unsigned fa = (unsigned)((wm * (BM / WM) + lrow) * 32 + (VY_SPLIT_SLOT(lrow, h) << 4));
  const unsigned fw = (unsigned)(W_BASE + (wn * (BN / WN) + lrow) * 32 + (VY_SPLIT_SLOT(lrow, h) << 4));

  auto phase = [&](auto xi_) {
    constexpr int XI = decltype(xi_)::value;
    // the two pixels of V_xi (dx relative to the centre of x0) and the sign of the second one
    constexpr int DX1 = XI == 0 ? -1 : (XI == 2 ? 1 : 0), DX2 = XI == 0 ? 1 : (XI == 1 ? 1 : (XI == 2 ? 0 : 2));
    constexpr bool ADD = XI == 1;
    const unsigned char* w_tile = a.wimg + XI * a.wimg_bytes + (long long)(n0 >> 5) * KS * 3072;
    const int T = KS;
    int n_tap = 0, n_cc = 0, a_koff = 0;
    long long w_koff = 0;
    auto advance = [&]() {
      a_koff = ((n_tap - 1) * Wp) * a.Cin + n_cc * 16;
      w_koff = (long long)(n_tap * cch + n_cc) * 3072;
      if (++n_cc == cch) {
        n_cc = 0;
        ++n_tap;
      }
    };
    f32x4 av[2], bv[2];
    auto load_a = [&]() {
      if (!a_active) return;
      const f32x4* p = reinterpret_cast<const f32x4*>(a_ptr + a_koff + DX1 * a.Cin);
      const f32x4* q = reinterpret_cast<const f32x4*>(a_ptr + a_koff + DX2 * a.Cin);
      av[0] = p[0];
      av[1] = p[1];
      bv[0] = q[0];
      bv[1] = q[1];
    };
    auto dma_w = [&](int stage) {
#pragma unroll
      for (int j = 0; j < W_INSTR; ++j)
        if (W_TOTAL % NW == 0 || j * NW + wave < W_TOTAL)
          lds_dma16_s(w_voff[j], reinterpret_cast<const float*>(w_tile + w_koff), lds0 + stage * W_ST + w_lds[j]);
    };
    auto store_a = [&](int stage) {
      if (!a_active) return;
      f32x4 v0, v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = ADD ? av[0][e] + bv[0][e] : av[0][e] - bv[0][e];
        v1[e] = ADD ? av[1][e] + bv[1][e] : av[1][e] - bv[1][e];
      }
      vy_u32x4 H, M, L;
      split8(v0, v1, H, M, L);
      unsigned char* d = smem + stage * A_ST + a_lds;
      *reinterpret_cast<vy_u32x4*>(d) = H;
      *reinterpret_cast<vy_u32x4*>(d + A_PL) = M;
      *reinterpret_cast<vy_u32x4*>(d + 2 * A_PL) = L;
    };
    auto compute = [&](const unsigned char* sa, const unsigned char* sw) {
      bf16x8 af[3][TMs], wf[3][TNs];
      auto rd_a = [&](int p) {
#pragma unroll
        for (int i = 0; i < TMs; ++i) af[p][i] = *reinterpret_cast<const bf16x8*>(sa + fa + p * A_PL + i * 1024);
      };
      auto rd_w = [&](int p) {
#pragma unroll
        for (int j = 0; j < TNs; ++j) wf[p][j] = *reinterpret_cast<const bf16x8*>(sw + fw + p * W_PL + j * 1024);
      };
      auto prod = [&](int pa, int pw) {
#pragma unroll
        for (int i = 0; i < TMs; ++i)
#pragma unroll
          for (int j = 0; j < TNs; ++j)
            acc[XI][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], wf[pw][j], acc[XI][i][j], 0, 0, 0);
      };
      rd_a(2);
      rd_w(0);
      rd_a(0);
      rd_w(2);
      rd_a(1);
      rd_w(1);
      prod(2, 0);
      prod(0, 2);
      prod(1, 1);
      prod(1, 0);
      prod(0, 1);
      prod(0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using Y = std::true_type;
    using N_ = std::false_type;
    auto kstep = [&](auto st_, auto has1_, auto has2_) {
      const int ST = st_;
      constexpr bool HAS1 = decltype(has1_)::value, HAS2 = decltype(has2_)::value;
      lds_barrier();
      if (HAS1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_a((ST + 1) % 3);
      }
      if (HAS2) {
        advance();
        load_a();
        dma_w((ST + 2) % 3);
      }
      compute(smem + ST * A_ST, smem + ST * W_ST);
    };
    lds_barrier();  // the previous phase's last k-step may still be reading stages 0 .. 2
    advance();
    load_a();
    dma_w(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_a(0);
    if (T > 1) {
      advance();
      load_a();
      dma_w(1);
    }
    int t = 0;
    for (; t + 5 <= T; t += 3) {
      kstep(S0{}, Y{}, Y{});
      kstep(S1{}, Y{}, Y{});
      kstep(S2{}, Y{}, Y{});
    }
    for (; t < T; ++t) {
      const int after = T - 1 - t, st = t % 3;
      if (after >= 2) kstep(st, Y{}, Y{});
      else if (after == 1) kstep(st, Y{}, N_{});
      else kstep(st, N_{}, N_{});
    }
  };
  phase(std::integral_constant<int, 0>{});
  phase(std::integral_constant<int, 1>{});
  phase(std::integral_constant<int, 2>{});
  phase(std::integral_constant<int, 3>{});

  // output transform + affine + leaky; C/D map: column lane & 31, row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < TNs; ++j) {
    const int n = n0 + wn * (BN / WN) + j * 32 + lrow;
    const float sc = a.scale[n], sh = a.shift[n];
#pragma unroll
    for (int i = 0; i < TMs; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long long o = o_off[row];
        if (o < 0) continue;
        const float m0v = acc[0][i][j][r], m1v = acc[1][i][j][r], m2v = acc[2][i][j][r], m3v = acc[3][i][j][r];
        const float y0 = (m0v + m1v) + m2v, y1 = (m1v - m2v) - m3v;
        a.out[o + n] = vy_leaky(fmaf(y0, sc, sh));
        const int xp = (m0 + row) % a.Wp2;
        if (2 * xp + 1 < a.W) a.out[o + a.Cout + n] = vy_leaky(fmaf(y1, sc, sh));
      }
  }
#endif
}

// ---- version 2: TWO xi per k-step (24 MFMAs per wave and barrier interval, like the product's 128 x 128 tile): block = 64
// pairs x 128 channels, two phases (xi 0,1 then 2,3), three pixel loads per k-step feed both V's, two LDS stages of
// [2 xi][3 planes] for A and for W (74 KB: two blocks per CU), W DMA and A loads one k-step ahead with a counted vmcnt.
__global__ __launch_bounds__(256, 2) void conv_wino2_kernel(const WinoArgs a, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 64, BN = 128, WN = 2, NW = 4, NT = 256;
  constexpr int A_PL = BM * 32, W_PL = BN * 32, A_XI = 3 * A_PL, W_XI = 3 * W_PL, A_ST = 2 * A_XI, W_ST = 2 * W_XI;
  constexpr int W_BASE = 2 * A_ST;
  constexpr int TNs = 2;  // wave tile: 32 pairs x 64 channels
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_ST + 2 * W_ST + BM * 16];
  long long* in_off = reinterpret_cast<long long*>(smem + 2 * A_ST + 2 * W_ST);
  long long* o_off = in_off + BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, lrow = lane & 31;
  const int cch = a.Cin >> 4;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Wp = a.W + 2;
  for (int rr = tid; rr < BM; rr += NT) {
    const int m = m0 + rr, mm = m < a.Mp ? m : a.Mp - 1;
    const int xp = mm % a.Wp2, t = mm / a.Wp2, y = t % a.H, b = t / a.H;
    const long long pix = ((long long)(b * (a.H + 2) + y + 1) * Wp + 2 * xp + 1);
    in_off[rr] = pix * a.Cin;
    o_off[rr] = m < a.Mp ? pix * a.Cout : -1;
  }
  __syncthreads();
  const bool a_active = wave < 2;  // 64 rows x 2 octets = 128 staging threads
  const int row_s = (tid & 127) >> 1, oct_s = tid & 1;
  const float* a_ptr = a.in + in_off[row_s] + oct_s * 8;
  const unsigned a_lds = (unsigned)(row_s * 32 + (VY_SPLIT_SLOT(row_s, oct_s) << 4));
  const int KS = 3 * cch;
  // W DMA: instruction q = j * 4 + wave of 24: image e = q / 12, row group g = (q % 12) / 3, plane p = q % 3
  unsigned w_voff[6], w_lds[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int q = (j * NW + wave) % 12, g = q / 3, p = q - g * 3;
    w_voff[j] = (unsigned)(g * KS * 3072 + p * 1024 + lane * 16);
    w_lds[j] = (unsigned)(W_BASE + (j >= 3 ? W_XI : 0) + p * W_PL + g * 1024);
  }
  f32x16 acc[4][TNs];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int j = 0; j < TNs; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][j][r] = 0.0f;
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const unsigned fa = (unsigned)((wm * 32 + lrow) * 32 + (VY_SPLIT_SLOT(lrow, h) << 4));
  const unsigned fw = (unsigned)(W_BASE + (wn * 64 + lrow) * 32 + (VY_SPLIT_SLOT(lrow, h) << 4));

  auto phase = [&](auto ph_) {
    constexpr int PH = decltype(ph_)::value;  // xi = 2 PH, 2 PH + 1
    constexpr int DXA = PH == 0 ? -1 : 0, DXB = PH == 0 ? 0 : 1, DXC = PH == 0 ? 1 : 2;  // d0 d1 d2  |  d1 d2 d3
    const unsigned char* w_tile0 = a.wimg + (2 * PH) * a.wimg_bytes + (long long)(n0 >> 5) * KS * 3072;
    const unsigned char* w_tile1 = w_tile0 + a.wimg_bytes;
    const int T = KS;
    int a_tap = 0, a_cc = 0, a_koff = 0, w_tap = 0, w_cc = 0;
    long long w_koff = 0;
    auto advance_a = [&]() {
      a_koff = ((a_tap - 1) * Wp) * a.Cin + a_cc * 16;
      if (++a_cc == cch) a_cc = 0, ++a_tap;
    };
    auto advance_w = [&]() {
      w_koff = (long long)(w_tap * cch + w_cc) * 3072;
      if (++w_cc == cch) w_cc = 0, ++w_tap;
    };
    f32x4 pa[2], pb[2], pc[2];
    auto load_a = [&]() {
      if (!a_active) return;
      const f32x4* x0 = reinterpret_cast<const f32x4*>(a_ptr + a_koff + DXA * a.Cin);
      const f32x4* x1 = reinterpret_cast<const f32x4*>(a_ptr + a_koff + DXB * a.Cin);
      const f32x4* x2 = reinterpret_cast<const f32x4*>(a_ptr + a_koff + DXC * a.Cin);
      pa[0] = x0[0], pa[1] = x0[1];
      pb[0] = x1[0], pb[1] = x1[1];
      pc[0] = x2[0], pc[1] = x2[1];
    };
    auto dma_w = [&](int stage) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        lds_dma16_s(w_voff[j], reinterpret_cast<const float*>((j >= 3 ? w_tile1 : w_tile0) + w_koff), lds0 + stage * W_ST + w_lds[j]);
    };
    auto store_a = [&](int stage) {
      if (!a_active) return;
      f32x4 u0, u1, v0, v1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (PH == 0) {  // V0 = d0 - d2, V1 = d1 + d2
          u0[e] = pa[0][e] - pc[0][e], u1[e] = pa[1][e] - pc[1][e];
          v0[e] = pb[0][e] + pc[0][e], v1[e] = pb[1][e] + pc[1][e];
        } else {        // V2 = d2 - d1, V3 = d1 - d3
          u0[e] = pb[0][e] - pa[0][e], u1[e] = pb[1][e] - pa[1][e];
          v0[e] = pa[0][e] - pc[0][e], v1[e] = pa[1][e] - pc[1][e];
        }
      }
      vy_u32x4 H, M, L;
      unsigned char* d = smem + stage * A_ST + a_lds;
      split8(u0, u1, H, M, L);
      *reinterpret_cast<vy_u32x4*>(d) = H;
      *reinterpret_cast<vy_u32x4*>(d + A_PL) = M;
      *reinterpret_cast<vy_u32x4*>(d + 2 * A_PL) = L;
      split8(v0, v1, H, M, L);
      *reinterpret_cast<vy_u32x4*>(d + A_XI) = H;
      *reinterpret_cast<vy_u32x4*>(d + A_XI + A_PL) = M;
      *reinterpret_cast<vy_u32x4*>(d + A_XI + 2 * A_PL) = L;
    };
    auto compute = [&](const unsigned char* sa, const unsigned char* sw) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        bf16x8 af[3], wf[3][TNs];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          af[p] = *reinterpret_cast<const bf16x8*>(sa + e * A_XI + fa + p * A_PL);
#pragma unroll
          for (int j = 0; j < TNs; ++j) wf[p][j] = *reinterpret_cast<const bf16x8*>(sw + e * W_XI + fw + p * W_PL + j * 1024);
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int j = 0; j < TNs; ++j)
            acc[2 * PH + e][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]], wf[PB[t]][j], acc[2 * PH + e][j], 0, 0, 0);
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using Y = std::true_type;
    using N_ = std::false_type;
    // k-step t on stage ST = t & 1: A(t + 1) is in registers, W(t + 1) not yet issued
    auto kstep = [&](auto st_, auto has1_, auto has2_) {
      const int ST = st_;
      constexpr bool HAS1 = decltype(has1_)::value, HAS2 = decltype(has2_)::value;
      lds_barrier();
      if (HAS1) {
        store_a(ST ^ 1);
        advance_w();
        dma_w(ST ^ 1);
      }
      if (HAS2) {
        advance_a();
        load_a();
      }
      compute(smem + ST * A_ST, smem + ST * W_ST);
      if (HAS1) {  // this wave's W(t + 1) DMA must have landed before the next barrier; the A(t + 2) loads are younger
        if (HAS2 && a_active) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    lds_barrier();  // the previous phase's last k-step may still be reading
    advance_a();
    load_a();
    advance_w();
    dma_w(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_a(0);
    if (T > 1) {
      advance_a();
      load_a();
    }
    int t = 0;
    for (; t + 4 <= T; t += 2) {
      kstep(S0{}, Y{}, Y{});
      kstep(S1{}, Y{}, Y{});
    }
    for (; t < T; ++t) {
      const int after = T - 1 - t;
      if (t & 1) {
        if (after >= 2) kstep(S1{}, Y{}, Y{});
        else if (after == 1) kstep(S1{}, Y{}, N_{});
        else kstep(S1{}, N_{}, N_{});
      } else {
        if (after >= 2) kstep(S0{}, Y{}, Y{});
        else if (after == 1) kstep(S0{}, Y{}, N_{});
        else kstep(S0{}, N_{}, N_{});
      }
    }
  };
  phase(std::integral_constant<int, 0>{});
  phase(std::integral_constant<int, 1>{});

#pragma unroll
  for (int j = 0; j < TNs; ++j) {
    const int n = n0 + wn * 64 + j * 32 + lrow;
    const float sc = a.scale[n], sh = a.shift[n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const long long o = o_off[row];
      if (o < 0) continue;
      const float m0v = acc[0][j][r], m1v = acc[1][j][r], m2v = acc[2][j][r], m3v = acc[3][j][r];
      const float y0 = (m0v + m1v) + m2v, y1 = (m1v - m2v) - m3v;
      a.out[o + n] = vy_leaky(fmaf(y0, sc, sh));
      const int xp = (m0 + row) % a.Wp2;
      if (2 * xp + 1 < a.W) a.out[o + a.Cout + n] = vy_leaky(fmaf(y1, sc, sh));
    }
  }
#endif
}

int main(int argc, char** argv) {
  if (argc < 5) return fprintf(stderr, "usage: %s B H Cin Cout [reps=20]\n", argv[0]), 2;
  const int B = atoi(argv[1]), H = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]);
  const int reps = argc > 5 ? atoi(argv[5]) : 20;
  const int W = H, Wp = W + 2;
  if (Cin % 32 || Cout % WINO_BN) return fprintf(stderr, "Cin %% 32, Cout %% %d\n", WINO_BN), 2;
  std::vector<float> h_in((size_t)B * (H + 2) * Wp * Cin, 0.f), h_w((size_t)Cout * 9 * Cin), h_sc(Cout), h_sh(Cout);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() {
    float acc = 0.f;
    for (int j = 0; j < 4; ++j) {
      st ^= st << 13;
      st ^= st >> 7;
      st ^= st << 17;
      acc += (float)(st >> 40) * (1.0f / 16777216.0f) - 0.5f;
    }
    return acc * 1.7320508f;
  };
  for (int b = 0; b < B; ++b)
    for (int y = 1; y <= H; ++y)
      for (int x = 1; x <= W; ++x)
        for (int c = 0; c < Cin; ++c) h_in[(((size_t)b * (H + 2) + y) * Wp + x) * Cin + c] = rnd();
  const float ws = sqrtf(2.0f / (9.0f * Cin));
  for (auto& v : h_w) v = rnd() * ws;  // [o][dy][dx][c]
  for (int o = 0; o < Cout; ++o) {
    h_sc[o] = 1.0f + 0.1f * rnd();
    h_sh[o] = 0.1f * rnd();
  }
  // transformed weights U[xi][o][dy][c]
  std::vector<float> h_u((size_t)4 * Cout * 3 * Cin);
  for (int o = 0; o < Cout; ++o)
    for (int dy = 0; dy < 3; ++dy)
      for (int c = 0; c < Cin; ++c) {
        const float g0 = h_w[(((size_t)o * 3 + dy) * 3 + 0) * Cin + c], g1 = h_w[(((size_t)o * 3 + dy) * 3 + 1) * Cin + c],
                    g2 = h_w[(((size_t)o * 3 + dy) * 3 + 2) * Cin + c];
        const size_t e = ((size_t)o * 3 + dy) * Cin + c, P = (size_t)Cout * 3 * Cin;
        h_u[0 * P + e] = g0;
        h_u[1 * P + e] = ((g0 + g1) + g2) * 0.5f;
        h_u[2 * P + e] = ((g0 - g1) + g2) * 0.5f;
        h_u[3 * P + e] = g2;
      }
  float *d_in, *d_out, *d_u, *d_sc, *d_sh;
  unsigned char* d_img;
  const size_t out_n = (size_t)B * (H + 2) * Wp * Cout, img_b = vy_split_weight_bytes(Cout, 3, Cin);
  CK(hipMalloc(&d_in, h_in.size() * 4));
  CK(hipMalloc(&d_out, out_n * 4));
  CK(hipMalloc(&d_u, h_u.size() * 4));
  CK(hipMalloc(&d_sc, Cout * 4));
  CK(hipMalloc(&d_sh, Cout * 4));
  CK(hipMalloc(&d_img, img_b * 4));
  CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_u, h_u.data(), h_u.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_sc, h_sc.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_sh, h_sh.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_out, 0, out_n * 4));
  for (int xi = 0; xi < 4; ++xi)
    CK(vy_launch_split_weights(d_u + (size_t)xi * Cout * 3 * Cin, d_img + xi * img_b, Cout, 3, Cin, 0));
  WinoArgs a;
  a.in = d_in; a.out = d_out; a.wimg = d_img; a.wimg_bytes = (long long)img_b; a.scale = d_sc; a.shift = d_sh;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Wp2 = (W + 1) / 2; a.Mp = B * H * a.Wp2;
  const int tiles_m = (a.Mp + WINO_BM - 1) / WINO_BM, tiles_n = Cout / WINO_BN;
  const bool v2 = getenv("WINO_V2") != nullptr;
  if (v2 && Cout % 128) return fprintf(stderr, "v2: Cout %% 128\n"), 2;
  const int tiles_m2 = (a.Mp + 63) / 64, tiles_n2 = Cout / 128;
  auto run = [&]() {
    if (v2) hipLaunchKernelGGL(conv_wino2_kernel, dim3(tiles_m2 * tiles_n2), dim3(256), 0, 0, a, tiles_n2);
    else hipLaunchKernelGGL(conv_wino_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, 0, a, tiles_n);
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int round = 0; round < 2; ++round) {
    for (int i = 0; i < 3; ++i) run();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double gflop = 2.0 * B * H * W * (double)Cout * 9.0 * Cin * 1e-9, us = ms * 1e3 / reps;
    printf("%s[%dx%d] wino F(2,3) B=%d H=%d Cin=%d Cout=%d | %d pairs x %d | %.1f us  %.1f TF-equivalent of the direct conv (%.0f TF bf16 actually issued)\n",
           v2 ? "v2 " : "", v2 ? 64 : WINO_BM, v2 ? 128 : WINO_BN, B, H, Cin, Cout, a.Mp, Cout, us, gflop / us * 1e3, gflop / us * 1e3 * 6.0 / 1.5);
  }
  std::vector<float> h_out(out_n);
  CK(hipMemcpy(h_out.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
  double err = 0, maxv = 0;
  unsigned long long s2 = 424242ull;
  for (int smp = 0; smp < 4096; ++smp) {
    s2 = s2 * 6364136223846793005ull + 1442695040888963407ull;
    const int b = (int)((s2 >> 33) % B);
    s2 = s2 * 6364136223846793005ull + 1442695040888963407ull;
    const int y = (int)((s2 >> 33) % H);
    s2 = s2 * 6364136223846793005ull + 1442695040888963407ull;
    const int x = (int)((s2 >> 33) % W);
    s2 = s2 * 6364136223846793005ull + 1442695040888963407ull;
    const int o = (int)((s2 >> 33) % Cout);
    double s = 0;
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx)
        for (int c = 0; c < Cin; ++c)
          s += (double)h_in[(((size_t)b * (H + 2) + y + dy) * Wp + x + dx) * Cin + c] * (double)h_w[(((size_t)o * 3 + dy) * 3 + dx) * Cin + c];
    double v = s * (double)h_sc[o] + (double)h_sh[o];
    v = v > 0.1 * v ? v : 0.1 * v;
    const double got = h_out[(((size_t)b * (H + 2) + y + 1) * Wp + x + 1) * Cout + o];
    err = fmax(err, fabs(got - v));
    maxv = fmax(maxv, fabs(v));
  }
  printf("  max error vs float64 on 4096 samples: %.3e (max |value| %.3f)\n", err, maxv);
  return 0;
}
