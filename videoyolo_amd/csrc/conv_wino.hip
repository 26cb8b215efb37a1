// conv_wino.hip — the split-fp32 conv (conv_split.hip; conv mode VY_CONV_SPLIT_BF16X3, inference) for the long-K 3x3
// stride-1 conv + BN + leaky cells as a 1-D Winograd F(2, 3) along x: a third fewer multiplications on the bf16 matrix
// core, whose power draw is what bounds conv_split.hip on those layers.  Opt-in like the rest of that mode, not the parity
// path; same reference operator chain (models/definitions/layers.py:63-70, residual add darknet/three_darknet.py:119-123).
//
// For an output pixel pair (x0 = 2 xp, x0 + 1) of row y, with d_j = the input pixel at x0 - 1 + j of row y + dy and
// g_k = the weight at (dy, dx = k):
//     V_0 = d0 - d2   V_1 = d1 + d2   V_2 = d2 - d1   V_3 = d1 - d3
//     U_0 = g0        U_1 = (g0 + g1 + g2) / 2        U_2 = (g0 - g1 + g2) / 2        U_3 = g2
//     M_xi = sum over (dy, channel) of V_xi U_xi            Y(x0) = M_0 + M_1 + M_2      Y(x0 + 1) = M_1 - M_2 - M_3
// — four GEMMs of K = 3 Cin instead of one of K = 9 Cin over half as many rows.  V is formed in registers from two
// pixels of the row the A load touches anyway (one V per thread: waves 0-1 the first xi of a phase, waves 2-3 the second —
// forming both in waves 0-1 from three loads left waves 2-3 idle while they cut: 2 % slower) and cut into the three bf16
// planes exactly as conv_split.hip cuts its activations (x = h + m + l, six products per multiply); U is transformed in fp32 and cut ONCE per parameter change
// into four tile images of the split kernel's format (wino_weights_kernel).  Rounding differs from the direct form
// (sums of two inputs are rounded to fp32 before they are cut; the output transform adds three fp32 numbers): measured
// against float64 it is as close as the exact fma chain (tools/probe/conv_wino_probe.hip: 1.7e-6 ... 5.4e-6).
//
// Block = 64 pairs x 128 channels, 4 waves (32 pairs x 64 channels each), four accumulator sets per wave (128 registers),
// two phases (xi = 0, 1 then 2, 3), TWO xi per k-step: 24 MFMAs per wave and barrier interval like the 128 x 128 tile of
// conv_split.hip.  LDS: two stages of [2 xi][3 planes] for A (64 rows) and W (128 rows), 74 KB: two blocks per CU.  A
// loads and W DMA run one k-step ahead; the wait before a barrier is counted (the four A loads are younger than the DMA).
// Measured beside the 128 x 128 direct tile at 608x608 batch 64 (profiles/r04_wino_probe.txt): 76x76 128->256 +9 %,
// 38x38 256->512 +16 %, 19x19 512->1024 +15 %; in the net (tools/layer_profile.py, VY_SPLIT_WINO=2 / 0) every supported cell
// wins from batch 8 on, none of a single frame's (those stay on conv_split.hip's k-split): vy_conv_wino_pays.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "../../include/vy_math.h"
#include "split_device.h"
#include "conv_cost_model.h"

// weights [cout][3][3][cin] fp32 -> four images (xi) of [cout / 32][3 cin / 16][plane][32 rows][2 slots][8 channels]
__global__ void wino_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, const int cout, const int cin,
                                    const long long img_elems) {
  const long long total = (long long)cout * 3 * cin;
  const int cch = cin >> 4, KS = 3 * cch;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % cin);
    const long long t_ = e / cin;
    const int dy = (int)(t_ % 3), n = (int)(t_ / 3);
    const float* g = w + ((long long)n * 9 + dy * 3) * cin + c;
    const float g0 = g[0], g1 = g[cin], g2 = g[2 * cin];
    const float u[4] = {g0, ((g0 + g1) + g2) * 0.5f, ((g0 - g1) + g2) * 0.5f, g2};
    auto rne = [](float f) -> unsigned {
      const unsigned v = __builtin_bit_cast(unsigned, f);
      return (v + 0x7fffu + ((v >> 16) & 1u)) >> 16;
    };
    const int ng = n >> 5, row = n & 31, ks = dy * cch + (c >> 4), oct = (c >> 3) & 1, j = c & 7;
    const long long base = (((long long)ng * KS + ks) * 3) * 512 + row * 16 + (VY_SPLIT_SLOT(row, oct) << 3) + j;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      const float x = u[xi];
      const unsigned h = rne(x);
      const float r = x - __builtin_bit_cast(float, h << 16);
      const unsigned m = rne(r);
      const float q = r - __builtin_bit_cast(float, m << 16);
      const unsigned l = rne(q);
      unsigned short* o = img + xi * img_elems + base;
      o[0] = (unsigned short)h;
      o[512] = (unsigned short)m;
      o[1024] = (unsigned short)l;
    }
  }
}

size_t vy_wino_weight_bytes(int cout, int cin) { return 4 * vy_split_weight_bytes(cout, 3, cin); }

hipError_t vy_launch_wino_weights(const float* w, void* img, int cout, int cin, hipStream_t s) {
  if (cout % 32 != 0 || cin % 16 != 0) return hipErrorInvalidValue;
  const long long total = (long long)cout * 3 * cin;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(wino_weights_kernel, dim3(blocks), dim3(256), 0, s, w, (unsigned short*)img, cout, cin,
                     (long long)(vy_split_weight_bytes(cout, 3, cin) / 2));
  return hipGetLastError();
}

// probe builds only (tools/probe/wino_abl_probe.hip): what each part of the k-loop costs.  Bits: 1 no A loads, 2 no cut,
// 4 no W DMA, 8 no MFMA, 16 no fragment reads.  Results are garbage with any bit set
#ifndef VY_WINO_ABL
#define VY_WINO_ABL 0
#endif

// BM = pairs per block: 64 (4 waves, two blocks per CU).  128 (8 waves, one block per CU: a W tile feeds twice the rows —
// half the LDS-DMA instructions and L2 bytes per MFMA) is instantiated by probe builds only (-DVY_WINO_BM128): measured
// equal within the run-to-run spread on every batch-64 shape and 15-45 % slower on short launches
// (profiles/r05_negative_results.txt section 1)
template <int BM>
__global__ __launch_bounds__(BM * 4, BM == 64 ? 2 : 1) void conv_wino_kernel(const ConvArgs a, const int tiles_n, const int Wp2,
                                                                              const int Mp, const long long wimg_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BN = 128, WN = 2, NW = BM / 16, NT = NW * 64, WJ = 24 / NW;
  constexpr int A_PL = BM * 32, W_PL = BN * 32, A_XI = 3 * A_PL, W_XI = 3 * W_PL, A_ST = 2 * A_XI, W_ST = 2 * W_XI;
  constexpr int W_BASE = 2 * A_ST;
  constexpr int TNs = 2;  // wave tile: 32 pairs x 64 channels
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_ST + 2 * W_ST + BM * 24];
  long long* in_off = reinterpret_cast<long long*>(smem + 2 * A_ST + 2 * W_ST);  // element offset of the centre pixel of x0
  // byte offsets of the pair's two pixels in `out` and in `res`, relative to the tile's first pixel (conv_split.hip's row
  // tables); kInvalidRow: no such pixel (rows past the last pair; x0 + 1 of an odd width) — the store / load is dropped
  unsigned* o_off0 = reinterpret_cast<unsigned*>(in_off + BM);
  unsigned* o_off1 = o_off0 + BM;
  unsigned* r_off0 = o_off1 + BM;
  unsigned* r_off1 = r_off0 + BM;
  constexpr unsigned kInvalidRow = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, lrow = lane & 31;
  const int cch = a.Kc >> 4;
  // XCD-aware order (conv_igemm.hip): blocks L, L+8, ... share an L2; contiguous run of tiles per XCD, n fastest
  int vblk;
  {
    const int nblk = gridDim.x, L = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, idx = L >> 3;
    vblk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = vblk / tiles_n, tile_n = vblk - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  long long pix0;
  {
    const int xp = m0 % Wp2, t = m0 / Wp2, y = t % a.LH, b = t / a.LH;
    pix0 = (long long)(b * a.o_Hp + y + a.o_oy) * a.o_Wp + 2 * xp + a.o_ox;
  }
  for (int rr = tid; rr < BM; rr += NT) {
    const int m = m0 + rr, mm = m < Mp ? m : Mp - 1;
    const int xp = mm % Wp2, t = mm / Wp2, y = t % a.LH, b = t / a.LH;
    in_off[rr] = ((long long)(b * a.a_Hp + y + a.a_oy) * a.a_Wp + 2 * xp + a.a_ox) * a.a_cs + a.a_co;
    const unsigned rel = (unsigned)(((long long)(b * a.o_Hp + y + a.o_oy) * a.o_Wp + 2 * xp + a.o_ox) - pix0);
    const bool ok0 = m < Mp, ok1 = ok0 && 2 * xp + 1 < a.LW;
    o_off0[rr] = ok0 ? rel * (unsigned)a.o_cs * 4u : kInvalidRow;
    o_off1[rr] = ok1 ? (rel + 1u) * (unsigned)a.o_cs * 4u : kInvalidRow;
    r_off0[rr] = ok0 ? rel * (unsigned)a.r_cs * 4u : kInvalidRow;
    r_off1[rr] = ok1 ? (rel + 1u) * (unsigned)a.r_cs * 4u : kInvalidRow;
  }
  __syncthreads();
  // A staging: 64 rows x 2 octets x 2 xi = 256 items, one per thread; waves 0-1 form the first xi of the phase's pair,
  // waves 2-3 the second (each V needs two of the row's pixels: four 16-B loads, one cut, three ds_writes per thread)
  const int row_s = (tid & (NT / 2 - 1)) >> 1, oct_s = tid & 1, e_s = wave / (NW / 2);
  const float* a_ptr = a.in + in_off[row_s] + oct_s * 8;
  // d3 of a pair without an x0 + 1 (odd width) would lie past the row's right border: it only feeds the discarded
  // Y(x0 + 1), so that row reads d1 again instead
  const int dx3 = o_off1[row_s] != kInvalidRow ? 2 * a.a_cs : 0;
  const unsigned a_lds = (unsigned)(row_s * 32 + (VY_SPLIT_SLOT(row_s, oct_s) << 4));
  const int KS = 3 * cch;
  // W DMA: instruction i = j * NW + wave of 24: image e = i / 12, row group g = (i % 12) / 3, plane p = i % 3
  unsigned w_voff[WJ], w_lds[WJ];
  bool w_img[WJ];  // wave-uniform
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int i = j * NW + wave, q = i % 12, g = q / 3, p = q - g * 3;
    w_img[j] = i >= 12;
    w_voff[j] = (unsigned)(g * KS * 3072 + p * 1024 + lane * 16);
    w_lds[j] = (unsigned)(W_BASE + (w_img[j] ? W_XI : 0) + p * W_PL + g * 1024);
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
  const int row_step = a.a_Wp * a.a_cs;

  auto phase = [&](auto ph_) {
    constexpr int PH = decltype(ph_)::value;  // xi = 2 PH, 2 PH + 1;  pixels d0 d1 d2  |  d1 d2 d3
    // V_0 = d0 - d2   V_1 = d1 + d2   V_2 = d2 - d1   V_3 = d1 - d3     (offsets of the minuend / first and the second pixel)
    const int dxp = PH == 0 ? (e_s == 0 ? -a.a_cs : 0) : (e_s == 0 ? a.a_cs : 0);
    const int dxq = PH == 0 ? a.a_cs : (e_s == 0 ? 0 : dx3);
    const float sg = PH == 0 && e_s == 1 ? 1.0f : -1.0f;
    const f32x4 sgn = {sg, sg, sg, sg};
    const unsigned char* w_tile0 =
        reinterpret_cast<const unsigned char*>(a.w_split) + (2 * PH) * wimg_bytes + (long long)(n0 >> 5) * KS * 3072;
    const unsigned char* w_tile1 = w_tile0 + wimg_bytes;
    const int T = KS;
    int a_tap = 0, a_cc = 0, a_koff = 0, w_tap = 0, w_cc = 0;
    long long w_koff = 0;
    auto advance_a = [&]() {
      a_koff = (a_tap - 1) * row_step + a_cc * 16;
      if (++a_cc == cch) a_cc = 0, ++a_tap;
    };
    auto advance_w = [&]() {
      w_koff = (long long)(w_tap * cch + w_cc) * 3072;
      if (++w_cc == cch) w_cc = 0, ++w_tap;
    };
    f32x4 pp[2], pq[2];
    auto load_a = [&]() {
      if (VY_WINO_ABL & 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(pp[0][e]), "+v"(pp[1][e]), "+v"(pq[0][e]), "+v"(pq[1][e]));
        return;
      }
      const f32x4* x0 = reinterpret_cast<const f32x4*>(a_ptr + a_koff + dxp);
      const f32x4* x1 = reinterpret_cast<const f32x4*>(a_ptr + a_koff + dxq);
      pp[0] = x0[0], pp[1] = x0[1];
      pq[0] = x1[0], pq[1] = x1[1];
    };
    auto dma_w = [&](int stage) {
      if (VY_WINO_ABL & 4) return;
#pragma unroll
      for (int j = 0; j < WJ; ++j)
        lds_dma16_s(w_voff[j], reinterpret_cast<const float*>((w_img[j] ? w_tile1 : w_tile0) + w_koff), lds0 + stage * W_ST + w_lds[j]);
    };
    auto store_a = [&](int stage) {
      if (VY_WINO_ABL & 2) {
        unsigned char* d = smem + stage * A_ST + e_s * A_XI + a_lds;
        *reinterpret_cast<f32x4*>(d) = pp[0];
        *reinterpret_cast<f32x4*>(d + A_PL) = pp[1];
        *reinterpret_cast<f32x4*>(d + 2 * A_PL) = pq[0];
        return;
      }
      // V = p +- q as fma(q, +-1, p): the product is exact, so this IS the rounded sum / difference — one (packed)
      // instruction per channel pair instead of an add, a subtract and a select on the wave-uniform `add`
      const f32x4 u0 = __builtin_elementwise_fma(pq[0], sgn, pp[0]), u1 = __builtin_elementwise_fma(pq[1], sgn, pp[1]);
      vy_u32x4 H, M, L;
      split8(u0, u1, H, M, L);
      unsigned char* d = smem + stage * A_ST + e_s * A_XI + a_lds;
      *reinterpret_cast<vy_u32x4*>(d) = H;
      *reinterpret_cast<vy_u32x4*>(d + A_PL) = M;
      *reinterpret_cast<vy_u32x4*>(d + 2 * A_PL) = L;
    };
    auto compute = [&](const unsigned char* sa, const unsigned char* sw) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        bf16x8 af[3], wf[3][TNs];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          if (VY_WINO_ABL & 16) {
            asm volatile("" : "=v"(af[p]));
#pragma unroll
            for (int j = 0; j < TNs; ++j) asm volatile("" : "=v"(wf[p][j]));
            continue;
          }
          af[p] = *reinterpret_cast<const bf16x8*>(sa + e * A_XI + fa + p * A_PL);
#pragma unroll
          for (int j = 0; j < TNs; ++j) wf[p][j] = *reinterpret_cast<const bf16x8*>(sw + e * W_XI + fw + p * W_PL + j * 1024);
        }
        if (VY_WINO_ABL & 8) {
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            asm volatile("" ::"v"(af[p]));
#pragma unroll
            for (int j = 0; j < TNs; ++j) asm volatile("" ::"v"(wf[p][j]));
          }
          continue;
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
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
      constexpr int ST = decltype(st_)::value;
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
      if (HAS1) {  // this wave's W(t + 1) DMA must have landed before the next barrier; the four A(t + 2) loads are younger
        if (HAS2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    lds_barrier();  // the previous phase's last k-step may still be reading
    if (VY_WINO_ABL & 1) pp[0] = pp[1] = pq[0] = pq[1] = f32x4{1.f, 2.f, 3.f, 4.f};
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

  // output transform, then the cell's epilogue (conv_igemm.hip): affine -> leaky -> + addend -> store, for both pixels,
  // through buffer descriptors based at the tile's first pixel (an offset with bit 31 set is out of range: dropped).
  // C/D map of the 32x32 MFMA: column lane & 31, row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  constexpr int kRsrcFlags = 0x00020000;
  const __amdgpu_buffer_rsrc_t out_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(a.out + (pix0 * a.o_cs + a.o_co + n0), 0, 0x7fffffff, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res ? a.res + (pix0 * a.r_cs + a.r_co + n0) : a.in), 0, 0x7fffffff, kRsrcFlags);
  auto epilogue = [&](auto has_res_) {
    constexpr bool has_res = decltype(has_res_)::value;
#pragma unroll
    for (int j = 0; j < TNs; ++j) {
      const int ncol = wn * 64 + j * 32 + lrow;
      const unsigned colc = (unsigned)ncol * 4u;
      const float sc = a.scale[n0 + ncol], sh = a.shift[n0 + ncol];
      unsigned oo0[16], oo1[16];
      float rv0[16], rv1[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        oo0[r] = __builtin_elementwise_add_sat(o_off0[row], colc);
        oo1[r] = __builtin_elementwise_add_sat(o_off1[row], colc);
        if (has_res) {
          rv0[r] = buf_load_f32(res_rsrc, __builtin_elementwise_add_sat(r_off0[row], colc));
          rv1[r] = buf_load_f32(res_rsrc, __builtin_elementwise_add_sat(r_off1[row], colc));
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float m0v = acc[0][j][r], m1v = acc[1][j][r], m2v = acc[2][j][r], m3v = acc[3][j][r];
        float y0 = vy_leaky(fmaf((m0v + m1v) + m2v, sc, sh)), y1 = vy_leaky(fmaf((m1v - m2v) - m3v, sc, sh));
        if (has_res) {
          y0 = y0 + rv0[r];
          y1 = y1 + rv1[r];
        }
        buf_store_f32(y0, out_rsrc, oo0[r], 0);
        buf_store_f32(y1, out_rsrc, oo1[r], 0);
      }
    }
  };
  if (a.res) epilogue(std::true_type{});
  else epilogue(std::false_type{});
#endif
}

bool vy_conv_wino_supported(const ConvArgs& a) {
  if (!a.w_wino || a.ntaps != 9 || a.a_s != 1 || a.ups != 1 || a.o_s != 1 || a.dgrad || a.stats) return false;
  if (!a.scale || !a.shift || !a.leaky) return false;  // the conv + BN + leaky cell (+ residual)
  if (a.Kc % 32 != 0 || a.N % 128 != 0 || a.LW < 2) return false;
  for (int t = 0; t < 9; ++t)
    if (a.tap_dy[t] != t / 3 - 1 || a.tap_dx[t] != t % 3 - 1 || a.tap_w[t] != t) return false;
  return true;
}

static long long wino_tiles(const ConvArgs& a, int bm) {
  const long long pairs = (long long)a.B * a.LH * ((a.LW + 1) / 2);
  return ((pairs + bm - 1) / bm) * (a.N / 128);
}

// pairs per block of the launch (probe builds: VY_WINO_BM=128)
static int wino_bm(const ConvArgs& a) {
#ifdef VY_WINO_BM128
  const char* f = getenv("VY_WINO_BM");
  if (f && atoi(f) == 128) return 128;
#endif
  return 64;
}

// Per launch, where the cost models say it wins (conv_cost_model.h: vy_predict_wino against the split kernel's and the
// exact kernel's predictions): at 608x608 every supported cell from batch 8 on, the 152x152 / 76x76 ones from batch 2, none of
// a single frame's (the split kernel's k-split covers those).  VY_SPLIT_WINO=0: never, =2: wherever supported (tests)
bool vy_conv_wino_pays(const ConvArgs& a) {
  if (!vy_conv_wino_supported(a)) return false;
  int mode = a.env_wino_mode_p1 - 1;  // read once per forward by the net; per call only for hand-made ConvArgs
  if (mode < 0) {
    const char* sw = getenv("VY_SPLIT_WINO");
    mode = sw ? atoi(sw) : 1;
  }
  if (mode == 0) return false;
  if (mode == 2) return true;
  const int cus = vy_args_cus(a);
  if (!vy_model_fitted(cus)) return false;  // fitted on 256 CUs (conv_cost_model.h)
  const long long pairs = (long long)a.B * a.LH * ((a.LW + 1) / 2);
  const double t_wino = vy_predict_wino(pairs, a.N, a.Kc, cus);
  double t_other = vy_conv_predict_us(a);
  if (vy_conv_split_supported(a)) {
    const long long max_ks = a.splitk_slabs ? std::max<long long>(1, (long long)(a.splitk_bytes / ((unsigned long long)a.M * a.N * 4ull))) : 1;
    int bm, bn, ks;
    t_other = std::min(t_other, vy_predict_split(a.M, a.N, 9.0 * a.Kc, (int)std::min<long long>(max_ks, 64), &bm, &bn, &ks, cus));
  }
  return t_wino < 0.97 * t_other;
}

hipError_t vy_launch_conv_wino(const ConvArgs& a, hipStream_t s) {
  if (!vy_conv_wino_supported(a) || a.LH < 1 || a.M <= 0) return hipErrorInvalidValue;
  const int Wp2 = (a.LW + 1) / 2, Mp = a.B * a.LH * Wp2, tiles_n = a.N / 128;
  ConvArgs k = a;
  k.w_split = a.w_wino;  // (the kernel reads its images through the same field)
  const int bm = wino_bm(a);
  const long long wb = (long long)vy_split_weight_bytes(a.w_cout, 3, a.Kc);
#ifdef VY_WINO_BM128
  if (bm == 128) {
    hipLaunchKernelGGL(conv_wino_kernel<128>, dim3((unsigned)wino_tiles(a, 128)), dim3(512), 0, s, k, tiles_n, Wp2, Mp, wb);
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL(conv_wino_kernel<64>, dim3((unsigned)wino_tiles(a, bm)), dim3(256), 0, s, k, tiles_n, Wp2, Mp, wb);
  return hipGetLastError();
}
