// PROTOTYPE (not product code): the split-fp32 3x3 stride-1 conv as a 1-D Winograd F(4, 3) along x — HALF the multiplications
// of the direct conv (the product's conv_wino.hip, F(2, 3): two thirds).  VERDICT r4 item 1, route (b).
//
// For an output quad (x0 = 4 xq ... x0 + 3) of row y, with d_j = the input pixel at x0 - 1 + j of row y + dy (j = 0..5) and
// g_k = the weight at (dy, dx = k):
//     V_0 = 4 d0 - 5 d2 + d4              U_0 = g0 / 4
//     V_1 = (d3 + d4) - 4 (d1 + d2)       U_1 = -(g0 + g1 + g2) / 6
//     V_2 = (d4 - d3) + 4 (d1 - d2)       U_2 = -(g0 - g1 + g2) / 6
//     V_3 = (d4 - d2) + 2 (d3 - d1)       U_3 = g0 / 24 + g1 / 12 + g2 / 6
//     V_4 = (d4 - d2) - 2 (d3 - d1)       U_4 = g0 / 24 - g1 / 12 + g2 / 6
//     V_5 = 4 d1 - 5 d3 + d5              U_5 = g2
//     M_xi = sum over (dy, channel) of V_xi U_xi
//     Y0 = M0 + M1 + M2 + M3 + M4     Y1 = (M1 - M2) + 2 (M3 - M4)     Y2 = (M1 + M2) + 4 (M3 + M4)     Y3 = (M1 - M2) + 8 (M3 - M4) + M5
// — six GEMMs of K = 3 Cin over a quarter as many rows.  Same arithmetic as conv_split.hip / conv_wino.hip underneath: V is
// formed in fp32 registers and cut into three bf16 planes, U is transformed (in double, rounded to fp32) and cut once; six
// bf16 products per multiply on v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//
// Block = 64 quads x 64 channels, 4 waves (32 quads x 32 channels each), SIX accumulator sets per wave (96 registers), two
// phases (xi = 0,1,2 then 3,4,5) with three xi per k-step: 18 MFMAs per wave and barrier interval.  A staging: one thread
// = 4 channels of one quad: five 16-B pixel loads per k-step feed the phase's three V's; W by LDS-DMA from six image sets of
// the split kernel's format.  LDS: two stages of [3 xi][3 planes] for A (64 rows) and W (64 rows): 72 KB, two blocks per CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-inline-asm \
//       -I include -o tools/probe/wino43_probe tools/probe/wino43_probe.hip videoyolo_amd/csrc/conv_small.hip
//   ./wino43_probe B H Cin Cout [res=0] [reps=40]
#include "../../videoyolo_amd/csrc/conv_igemm.hip"
#include "../../videoyolo_amd/csrc/conv_split.hip"

#include <algorithm>
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

struct W43Args {
  const float* in;            // [B][H+2][W+2][Cin], zero border
  float* out;                 // [B][H+2][W+2][Cout]
  const float* res;           // nullable, pixel map of out
  const unsigned char* wimg;  // 6 images (xi) of vy_split_weight_bytes(Cout, 3, Cin) bytes each
  long long wimg_bytes;
  const float *scale, *shift;
  int B, H, W, Cin, Cout, Wq, Mq;  // Wq = ceil(W / 4), Mq = B * H * Wq quads
};

// weights [cout][3][3][cin] fp32 -> six images (xi) of [cout / 32][3 cin / 16][plane][32 rows][2 slots][8 channels]
__global__ void wino43_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, const int cout, const int cin,
                                      const long long img_elems) {
  const long long total = (long long)cout * 3 * cin;
  const int cch = cin >> 4, KS = 3 * cch;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % cin);
    const long long t_ = e / cin;
    const int dy = (int)(t_ % 3), n = (int)(t_ / 3);
    const float* g = w + ((long long)n * 9 + dy * 3) * cin + c;
    const double g0 = g[0], g1 = g[cin], g2 = g[2 * cin];
    const float u[6] = {(float)(g0 / 4.0),
                        (float)(-(g0 + g1 + g2) / 6.0),
                        (float)(-(g0 - g1 + g2) / 6.0),
                        (float)(g0 / 24.0 + g1 / 12.0 + g2 / 6.0),
                        (float)(g0 / 24.0 - g1 / 12.0 + g2 / 6.0),
                        (float)g2};
    auto rne = [](float f) -> unsigned {
      const unsigned v = __builtin_bit_cast(unsigned, f);
      return (v + 0x7fffu + ((v >> 16) & 1u)) >> 16;
    };
    const int ng = n >> 5, row = n & 31, ks = dy * cch + (c >> 4), oct = (c >> 3) & 1, j = c & 7;
    const long long base = (((long long)ng * KS + ks) * 3) * 512 + row * 16 + (VY_SPLIT_SLOT(row, oct) << 3) + j;
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) {
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

#if defined(__HIP_DEVICE_COMPILE__)
// 4 consecutive channels -> the three bf16 planes (2 dwords each); split_device.h's cut
__device__ __forceinline__ void split4(const f32x4 v, unsigned (&H)[2], unsigned (&M)[2], unsigned (&L)[2]) {
  const vy_f32x2 x[2] = {{v[0], v[1]}, {v[2], v[3]}};
#pragma unroll
  for (int j = 0; j < 2; ++j) {
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

__global__ __launch_bounds__(256, 2) void conv_wino43_kernel(const W43Args a, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 64, BN = 64, NW = 4, NT = 256, NXI = 3;  // three xi per k-step
  constexpr int A_PL = BM * 32, W_PL = BN * 32, A_XI = 3 * A_PL, W_XI = 3 * W_PL, A_ST = NXI * A_XI, W_ST = NXI * W_XI;
  constexpr int W_BASE = 2 * A_ST;
  constexpr int W_PIECES = NXI * 3 * (BN / 32), WJ = (W_PIECES + NW - 1) / NW;  // 18 pieces of 1 KiB, 5 / 5 / 4 / 4 per wave
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_ST + 2 * W_ST + BM * 20];
  long long* in_off = reinterpret_cast<long long*>(smem + 2 * A_ST + 2 * W_ST);  // element offset of pixel x0 (= d1), channel 0
  long long* o_off = in_off + BM;                                               // element offset of output pixel x0
  int* nvalid = reinterpret_cast<int*>(o_off + BM);                              // pixels of the quad inside the row (0: dead row)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, lrow = lane & 31;
  const int cch = a.Cin >> 4, KS = 3 * cch;
  int vblk;
  {
    const int nblk = gridDim.x, L = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, idx = L >> 3;
    vblk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = vblk / tiles_n, tile_n = vblk - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int Wp = a.W + 2;
  for (int rr = tid; rr < BM; rr += NT) {
    const int m = m0 + rr, mm = m < a.Mq ? m : a.Mq - 1;
    const int xq = mm % a.Wq, t = mm / a.Wq, y = t % a.H, b = t / a.H;
    const long long pix = (long long)(b * (a.H + 2) + y + 1) * Wp + 4 * xq + 1;
    in_off[rr] = pix * a.Cin;
    o_off[rr] = pix * a.Cout;
    nvalid[rr] = m < a.Mq ? (a.W - 4 * xq < 4 ? a.W - 4 * xq : 4) : 0;
  }
  __syncthreads();
  // A staging: thread = (quad, 4-channel group of the 16-channel k-step)
  const int quad_s = tid >> 2, cq = tid & 3;
  const float* a_ptr = a.in + in_off[quad_s] + cq * 4;
  const unsigned a_lds = (unsigned)(quad_s * 32 + (VY_SPLIT_SLOT(quad_s, cq >> 1) << 4) + (cq & 1) * 8);
  // W DMA: piece i = j * 4 + wave of 18: image e = i / 6, row group g = (i % 6) / 3, plane p = i % 3
  unsigned w_voff[WJ], w_lds[WJ];
  int w_e[WJ];
  bool w_on[WJ];
#pragma unroll
  for (int j = 0; j < WJ; ++j) {
    const int i = j * NW + wave;
    w_on[j] = i < W_PIECES;
    const int ii = w_on[j] ? i : 0, e = ii / 6, r = ii - e * 6, g = r / 3, p = r - g * 3;
    w_e[j] = e;
    w_voff[j] = (unsigned)(g * KS * 3072 + p * 1024 + lane * 16);
    w_lds[j] = (unsigned)(W_BASE + e * W_XI + p * W_PL + g * 1024);
  }
  f32x16 acc[6];
#pragma unroll
  for (int x = 0; x < 6; ++x)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[x][r] = 0.0f;
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const unsigned fa = (unsigned)((wm * 32 + lrow) * 32 + (VY_SPLIT_SLOT(lrow, h) << 4));
  const unsigned fw = (unsigned)(W_BASE + (wn * 32 + lrow) * 32 + (VY_SPLIT_SLOT(lrow, h) << 4));
  const int row_step = Wp * a.Cin;

  auto phase = [&](auto ph_) {
    constexpr int PH = decltype(ph_)::value;  // xi = 3 PH + e;  pixels d0..d4 | d1..d5
    const unsigned char* w_tile = a.wimg + (long long)(3 * PH) * a.wimg_bytes + (long long)(n0 >> 5) * KS * 3072;
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
    f32x4 px[5];
    auto load_a = [&]() {
#pragma unroll
      for (int j = 0; j < 5; ++j) px[j] = *reinterpret_cast<const f32x4*>(a_ptr + a_koff + (j + PH - 1) * a.Cin);
    };
    auto dma_w = [&](int stage) {
#pragma unroll
      for (int j = 0; j < WJ; ++j)
        if (w_on[j])
          lds_dma16_s(w_voff[j], reinterpret_cast<const float*>(w_tile + (long long)w_e[j] * a.wimg_bytes + w_koff),
                      lds0 + stage * W_ST + w_lds[j]);
    };
    auto store_a = [&](int stage) {
      f32x4 v[3];
      if (PH == 0) {  // px = d0 d1 d2 d3 d4
        const f32x4 four = {4.f, 4.f, 4.f, 4.f}, m4 = {-4.f, -4.f, -4.f, -4.f}, m5 = {-5.f, -5.f, -5.f, -5.f};
        v[0] = __builtin_elementwise_fma(four, px[0], __builtin_elementwise_fma(m5, px[2], px[4]));
        v[1] = __builtin_elementwise_fma(m4, px[1] + px[2], px[3] + px[4]);
        v[2] = __builtin_elementwise_fma(four, px[1] - px[2], px[4] - px[3]);
      } else {        // px = d1 d2 d3 d4 d5
        const f32x4 two = {2.f, 2.f, 2.f, 2.f}, m2 = {-2.f, -2.f, -2.f, -2.f}, four = {4.f, 4.f, 4.f, 4.f}, m5 = {-5.f, -5.f, -5.f, -5.f};
        const f32x4 f = px[3] - px[1], g = px[2] - px[0];
        v[0] = __builtin_elementwise_fma(two, g, f);
        v[1] = __builtin_elementwise_fma(m2, g, f);
        v[2] = __builtin_elementwise_fma(four, px[0], __builtin_elementwise_fma(m5, px[2], px[4]));
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        unsigned H[2], M[2], L[2];
        split4(v[e], H, M, L);
        unsigned char* d = smem + stage * A_ST + e * A_XI + a_lds;
        *reinterpret_cast<uint2*>(d) = make_uint2(H[0], H[1]);
        *reinterpret_cast<uint2*>(d + A_PL) = make_uint2(M[0], M[1]);
        *reinterpret_cast<uint2*>(d + 2 * A_PL) = make_uint2(L[0], L[1]);
      }
    };
    auto compute = [&](const unsigned char* sa, const unsigned char* sw) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        bf16x8 af[3], wf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          af[p] = *reinterpret_cast<const bf16x8*>(sa + e * A_XI + fa + p * A_PL);
          wf[p] = *reinterpret_cast<const bf16x8*>(sw + e * W_XI + fw + p * W_PL);
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
#pragma unroll
        for (int t = 0; t < 6; ++t) acc[3 * PH + e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]], wf[PB[t]], acc[3 * PH + e], 0, 0, 0);
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using Y = std::true_type;
    using N_ = std::false_type;
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
      if (HAS1) {  // this wave's W(t + 1) DMA must have landed before the next barrier; the five A(t + 2) loads are younger
        if (HAS2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    };
    lds_barrier();
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

  // output transform, then affine -> leaky -> (+ addend) -> store for the quad's pixels.
  // C/D map of the 32x32 MFMA: column lane & 31, row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const int n = n0 + wn * 32 + lrow;
  const float sc = a.scale[n], sh = a.shift[n];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
    const int nv = nvalid[row];
    if (nv == 0) continue;
    const long long o = o_off[row] + n;
    const float m0v = acc[0][r], m1v = acc[1][r], m2v = acc[2][r], m3v = acc[3][r], m4v = acc[4][r], m5v = acc[5][r];
    const float s12 = m1v + m2v, d12 = m1v - m2v, s34 = m3v + m4v, d34 = m3v - m4v;
    float y[4];
    y[0] = (m0v + s12) + s34;
    y[1] = fmaf(2.0f, d34, d12);
    y[2] = fmaf(4.0f, s34, s12);
    y[3] = fmaf(8.0f, d34, d12) + m5v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= nv) break;
      float v = vy_leaky(fmaf(y[j], sc, sh));
      if (a.res) v = v + a.res[o + (long long)j * a.Cout];
      a.out[o + (long long)j * a.Cout] = v;
    }
  }
#endif
}

static void fill_normal(std::vector<float>& h, float scale, unsigned long long seed) {
  unsigned long long st = seed;
  for (size_t i = 0; i < h.size(); ++i) {
    float acc = 0.f;
    for (int j = 0; j < 4; ++j) {
      st ^= st << 13;
      st ^= st >> 7;
      st ^= st << 17;
      acc += (float)(st >> 40) * (1.0f / 16777216.0f) - 0.5f;
    }
    h[i] = acc * 1.7320508f * scale;
  }
}

int main(int argc, char** argv) {
  if (argc < 5) return fprintf(stderr, "usage: %s B H Cin Cout [res=0] [reps=40]\n", argv[0]), 2;
  const int B = atoi(argv[1]), H = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]);
  const int res = argc > 5 ? atoi(argv[5]) : 0, reps = argc > 6 ? atoi(argv[6]) : 40;
  if (Cin % 32 || Cout % 64) return fprintf(stderr, "Cin %% 32, Cout %% 64\n"), 2;
  const int W = H, Wp = H + 2;
  const size_t in_n = (size_t)B * Wp * Wp * Cin, out_n = (size_t)B * Wp * Wp * Cout, w_n = (size_t)Cout * 9 * Cin;
  std::vector<float> h_in(in_n), h_w(w_n), h_sc(Cout), h_sh(Cout), h_rs(res ? out_n : 0);
  fill_normal(h_in, 1.0f, 88172645463325252ull);
  fill_normal(h_w, 1.0f / sqrtf(9.0f * Cin), 1234567ull);
  fill_normal(h_sc, 0.2f, 99ull);
  for (auto& v : h_sc) v += 1.0f;
  fill_normal(h_sh, 0.5f, 777ull);
  if (res) fill_normal(h_rs, 1.0f, 4242ull);
  if (getenv("VY_PROBE_ZERO")) {
    std::fill(h_in.begin(), h_in.end(), 0.f);
    std::fill(h_w.begin(), h_w.end(), 0.f);
  }
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < Wp; ++y)
      for (int x = 0; x < Wp; ++x)
        if (y == 0 || x == 0 || y == H + 1 || x == H + 1) memset(&h_in[(((size_t)b * Wp + y) * Wp + x) * Cin], 0, Cin * 4);
  float *in, *out, *w, *sc, *sh, *rs = nullptr;
  unsigned char* wimg;
  const size_t img_b = vy_split_weight_bytes(Cout, 3, Cin);
  CK(hipMalloc(&in, in_n * 4 + 4096));  // (a quad at the right edge of the last row may read a few pixels past its row)
  CK(hipMalloc(&out, out_n * 4));
  CK(hipMalloc(&w, w_n * 4));
  CK(hipMalloc(&sc, Cout * 4));
  CK(hipMalloc(&sh, Cout * 4));
  CK(hipMalloc(&wimg, 6 * img_b));
  CK(hipMemset(in, 0, in_n * 4 + 4096));
  CK(hipMemcpy(in, h_in.data(), in_n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h_w.data(), w_n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, h_sc.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, h_sh.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemset(out, 0, out_n * 4));
  if (res) {
    CK(hipMalloc(&rs, out_n * 4));
    CK(hipMemcpy(rs, h_rs.data(), out_n * 4, hipMemcpyHostToDevice));
  }
  {
    const long long total = (long long)Cout * 3 * Cin;
    hipLaunchKernelGGL(wino43_weights_kernel, dim3((unsigned)std::min<long long>((total + 255) / 256, 8192)), dim3(256), 0, 0, w,
                       (unsigned short*)wimg, Cout, Cin, (long long)(img_b / 2));
    CK(hipGetLastError());
  }
  W43Args a;
  a.in = in; a.out = out; a.res = rs; a.wimg = wimg; a.wimg_bytes = (long long)img_b; a.scale = sc; a.shift = sh;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.Wq = (W + 3) / 4; a.Mq = B * H * a.Wq;
  const int tiles_m = (a.Mq + 63) / 64, tiles_n = Cout / 64;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  double best = 1e30;
  for (int round = 0; round < 3; ++round) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(conv_wino43_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, 0, a, tiles_n);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(conv_wino43_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, 0, a, tiles_n);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fmin(best, ms * 1e3 / reps);
  }
  const double gflop = 2.0 * B * H * H * (double)Cout * 9.0 * Cin * 1e-9;
  printf("F(4,3) wino B=%d H=%d Cin=%d Cout=%d res=%d | %.1f us  %.1f TF-eq (%.0f TF bf16 issued)\n", B, H, Cin, Cout, res, best,
         gflop / best * 1e3, gflop / best * 1e3 * 3.0);
  std::vector<float> h_out(out_n);
  CK(hipMemcpy(h_out.data(), out, out_n * 4, hipMemcpyDeviceToHost));
  double err = 0, maxv = 0;
  unsigned long long s2 = 424242ull;
  auto nxt = [&](int mod) {
    s2 = s2 * 6364136223846793005ull + 1442695040888963407ull;
    return (int)((s2 >> 33) % mod);
  };
  for (int smp = 0; smp < 4096; ++smp) {
    const int b = nxt(B), y = nxt(H), x = smp % 7 == 0 ? H - 1 - (smp % 3) : nxt(H), o = nxt(Cout);
    double s = 0;
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx)
        for (int c = 0; c < Cin; ++c)
          s += (double)h_in[(((size_t)b * Wp + y + dy) * Wp + x + dx) * Cin + c] * (double)h_w[(((size_t)o * 3 + dy) * 3 + dx) * Cin + c];
    double v = s * (double)h_sc[o] + (double)h_sh[o];
    v = v > 0.1 * v ? v : 0.1 * v;
    const size_t oi = (((size_t)b * Wp + y + 1) * Wp + x + 1) * Cout + o;
    if (res) v += h_rs[oi];
    err = fmax(err, fabs((double)h_out[oi] - v));
    maxv = fmax(maxv, fabs(v));
  }
  double border = 0;
  for (int b = 0; b < B; b += (B > 4 ? B / 4 : 1))
    for (int x = 0; x < Wp; ++x)
      for (int c = 0; c < Cout; c += 7) {
        border = fmax(border, fabs((double)h_out[(((size_t)b * Wp + 0) * Wp + x) * Cout + c]));
        border = fmax(border, fabs((double)h_out[(((size_t)b * Wp + H + 1) * Wp + x) * Cout + c]));
        border = fmax(border, fabs((double)h_out[(((size_t)b * Wp + x) * Wp + 0) * Cout + c]));
        border = fmax(border, fabs((double)h_out[(((size_t)b * Wp + x) * Wp + H + 1) * Cout + c]));
      }
  printf("  max error vs float64 on 4096 samples: %.3e (max |value| %.3f); border max %.1e\n", err, maxv, border);
  return err < 1e-4 && border == 0 ? 0 : 1;
}
