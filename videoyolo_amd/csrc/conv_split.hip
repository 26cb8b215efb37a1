// conv_split.hip — OPT-IN "split-fp32" forward convolution on the gfx950 bf16 matrix core (bf16 x 3, six partial
// products, fp32 accumulate).  Not the parity path: conv_igemm.hip (exact fp32 fma chains) stays the default and
// the headline; this instance is selected per net with vy_net_set_conv_mode(net, VY_CONV_SPLIT_BF16X3) and is
// reported separately by bench.py.
//
// Replaces the same reference operator chain as conv_igemm.hip (models/definitions/layers.py:63-70: Conv2D ->
// BatchNorm -> LeakyReLU(0.1); residual add darknet/three_darknet.py:119-123; `_upsample` + concat
// layers.py:11-20, yolo3.py:1167-1177) for the layers whose contraction is long enough to be matrix-bound.
// SURVEY 7 hard-part (iii) names split-fp32 as admissible where 1e-4 agreement is kept.
//
// Arithmetic.  Every fp32 operand x is cut into three bf16 numbers, round-to-nearest-even each time:
//     h = bf16(x),  m = bf16(x - h),  l = bf16(x - h - m)          (x - h and x - h - m are exact in fp32)
// so x = h + m + l EXACTLY (8 + 8 + 8 significand bits, |m| <= 2^-9 |x|, |l| <= 2^-17 |x|).  A product x*w is
// evaluated as  l_x h_w + h_x l_w + m_x m_w + m_x h_w + h_x m_w + h_x h_w  — six v_mfma_f32_32x32x16_bf16 per
// 16 channels, each bf16 x bf16 product exact, accumulated in fp32; the three products left out (m l, l m, l l)
// are below 2^-25 |x w|, i.e. below the rounding of the fp32 accumulation itself.  The result is NOT bit-equal to
// the exact chain (different summation tree inside the matrix core); tests/test_gpu_split.py holds the tolerances.
//
// Data flow.  Activation planes stay fp32 (kernels.h) — the split happens in registers on the way into LDS, so
// every other kernel of the net (stem, 1x1 convs on the exact kernel, decode) is shared with the exact path:
//   A (pixels): global_load_dwordx4 x2 per thread and k-step (8 channels of one pixel) -> 3 x 4 packed dwords
//               (11 vector instructions per channel pair, beside the bf16 MFMAs, which do not use the vector
//               pipe's FMA hardware the way the fp32 MFMA does) -> ds_write_b128 x3
//   W:          pre-split ONCE per parameter change by split_weights_kernel into the exact LDS tile images
//               [cout / 32][k-step][plane][32 rows][16 channels], so a k-step of a 128-channel tile is twelve
//               1-KiB LDS-DMA instructions (global_load_lds_dwordx4) reading consecutive memory
//   LDS stage:  [plane][row][2 x 16 B], the 16-B slot XOR-ed with bit 3 of the row: conflict-free ds_read_b128
// Block = 4 waves (2 x 2), tile 128 pixels x 128 channels, k-step 16 channels, three LDS stages (72 KiB: two
// blocks per CU, so one block's epilogue runs in the shadow of the other's k-loop).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "../../include/vy_math.h"
#include "conv_device.h"
#include "conv_cost_model.h"

#include "split_device.h"

// fp32 weights [cout][taps][cin] -> bf16 tile images: image (ng, ks, p) is 1 KiB = [32 rows][2 slots][8 channels],
// row r = cout % 32, k-step ks = tap * (cin / 16) + cin / 16 index, slot s holds channel octet s ^ ((r >> 3) & 1)
__global__ void split_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, const int cout,
                                     const int taps, const int cin) {
  const long long total = (long long)cout * taps * cin;
  const int KS = taps * (cin >> 4);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % cin);
    const long long t_ = e / cin;
    const int tap = (int)(t_ % taps), n = (int)(t_ / taps);
    const float x = w[e];
    auto rne = [](float f) -> unsigned {
      const unsigned u = __builtin_bit_cast(unsigned, f);
      return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    const unsigned h = rne(x);
    const float r = x - __builtin_bit_cast(float, h << 16);
    const unsigned m = rne(r);
    const float q = r - __builtin_bit_cast(float, m << 16);
    const unsigned l = rne(q);
    const int ng = n >> 5, row = n & 31, ks = tap * (cin >> 4) + (c >> 4), oct = (c >> 3) & 1, j = c & 7;
    const long long base = (((long long)ng * KS + ks) * 3) * 512 + row * 16 + (VY_SPLIT_SLOT(row, oct) << 3) + j;
    img[base] = (unsigned short)h;
    img[base + 512] = (unsigned short)m;
    img[base + 1024] = (unsigned short)l;
  }
}

// the same weights as the [k = cout][n = cin] operand of the data gradient (train_yolov3.py:631 through Convolution's
// backward): image rows = input channels, k-steps = (tap, 16 output channels), cout zero-padded to a multiple of 32
// (the prediction convs: 75 -> 96; the dz planes' padding channels are zero as well)
__global__ void split_weights_dgrad_kernel(const float* __restrict__ w, unsigned short* __restrict__ img, const int cout,
                                           const int taps, const int cin) {
  const int coutp = (cout + 31) & ~31;
  const long long total = (long long)coutp * taps * cin;
  const int KS = taps * (coutp >> 4);
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(e % cin);
    const long long t_ = e / cin;
    const int tap = (int)(t_ % taps), o = (int)(t_ / taps);
    const float x = o < cout ? w[e] : 0.0f;
    auto rne = [](float f) -> unsigned {
      const unsigned u = __builtin_bit_cast(unsigned, f);
      return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    const unsigned h = rne(x);
    const float r = x - __builtin_bit_cast(float, h << 16);
    const unsigned m = rne(r);
    const float q = r - __builtin_bit_cast(float, m << 16);
    const unsigned l = rne(q);
    const int ng = c >> 5, row = c & 31, ks = tap * (coutp >> 4) + (o >> 4), oct = (o >> 3) & 1, j = o & 7;
    const long long base = (((long long)ng * KS + ks) * 3) * 512 + row * 16 + (VY_SPLIT_SLOT(row, oct) << 3) + j;
    img[base] = (unsigned short)h;
    img[base + 512] = (unsigned short)m;
    img[base + 1024] = (unsigned short)l;
  }
}

size_t vy_split_weight_dgrad_bytes(int cout, int taps, int cin) {
  return (size_t)((cin + 31) / 32) * taps * (((cout + 31) & ~31) / 16) * 3072;
}

hipError_t vy_launch_split_weights_dgrad(const float* w, void* img, int cout, int taps, int cin, hipStream_t s) {
  if (cin % 32 != 0) return hipErrorInvalidValue;
  const long long total = (long long)((cout + 31) & ~31) * taps * cin;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(split_weights_dgrad_kernel, dim3(blocks), dim3(256), 0, s, w, (unsigned short*)img, cout, taps, cin);
  return hipGetLastError();
}

// Every conv's images in ONE launch (the per-conv launches above cost 70 x 2 kernel boundaries per training step):
// descriptors sorted by `first` (prefix sum of the OCTET counts); a thread finds its conv by bisection and converts one
// 8-channel octet: 32 B in (forward images: contiguous; data-gradient images: 8 output channels at one (tap, cin), each
// read coalesced across the threads of a wave, which differ in cin), three 16-B stores out.
__global__ __launch_bounds__(256) void split_weights_batch_kernel(const float* __restrict__ params, unsigned char* __restrict__ ws,
                                                                  const SplitDesc* __restrict__ d, const int n,
                                                                  const long long total) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (d[mid].first <= g) lo = mid;
      else hi = mid - 1;
    }
    const SplitDesc c = d[lo];
    const long long e = g - c.first;  // octet index inside this image set
    const int cin = c.cin, taps = c.taps;
    float x[8];
    long long base;
    if (!c.dgrad) {  // octet = 8 consecutive input channels of (o, tap)
      const int co = cin >> 3;
      const int oc = (int)(e % co);
      const long long t_ = e / co;
      const int tap = (int)(t_ % taps), o = (int)(t_ / taps);
      const f32x4* src = reinterpret_cast<const f32x4*>(params + c.w_off + ((long long)o * taps + tap) * cin + oc * 8);
      const f32x4 v0 = src[0], v1 = src[1];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = v0[j];
        x[4 + j] = v1[j];
      }
      const int KS = taps * (cin >> 4);
      const int ng = o >> 5, row = o & 31, ks = tap * (cin >> 4) + (oc >> 1), oct = oc & 1;
      base = (((long long)ng * KS + ks) * 3) * 1024 + row * 32 + (VY_SPLIT_SLOT(row, oct) << 4);
    } else {  // octet = 8 consecutive output channels at (tap, cin); cout zero-padded to a multiple of 32
      const int coutp = (c.cout + 31) & ~31;
      const int ci = (int)(e % cin);
      const long long t_ = e / cin;
      const int tap = (int)(t_ % taps), oo = (int)(t_ / taps);  // output-channel octet
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int o = oo * 8 + j;
        x[j] = o < c.cout ? params[c.w_off + ((long long)o * taps + tap) * cin + ci] : 0.0f;
      }
      const int KS = taps * (coutp >> 4);
      const int ng = ci >> 5, row = ci & 31, ks = tap * (coutp >> 4) + (oo >> 1), oct = oo & 1;
      base = (((long long)ng * KS + ks) * 3) * 1024 + row * 32 + (VY_SPLIT_SLOT(row, oct) << 4);
    }
    auto rne = [](float f) -> unsigned {
      const unsigned u = __builtin_bit_cast(unsigned, f);
      return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    };
    vy_u32x4 H, M, L;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned hh[2], mm[2], ll[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float v = x[2 * j + q];
        hh[q] = rne(v);
        const float r = v - __builtin_bit_cast(float, hh[q] << 16);
        mm[q] = rne(r);
        ll[q] = rne(r - __builtin_bit_cast(float, mm[q] << 16));
      }
      H[j] = hh[0] | (hh[1] << 16);
      M[j] = mm[0] | (mm[1] << 16);
      L[j] = ll[0] | (ll[1] << 16);
    }
    unsigned char* img = ws + c.img_off + base;  // byte offset: image (ng, ks, plane) = 1 KiB, row 32 B, slot 16 B
    *reinterpret_cast<vy_u32x4*>(img) = H;
    *reinterpret_cast<vy_u32x4*>(img + 1024) = M;
    *reinterpret_cast<vy_u32x4*>(img + 2048) = L;
  }
#endif
}

hipError_t vy_launch_split_weights_batch(const float* params, void* ws, const SplitDesc* descs_dev, int n, long long total,
                                         hipStream_t s) {
  if (n < 1 || total < 1) return hipSuccess;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(split_weights_batch_kernel, dim3(blocks), dim3(256), 0, s, params, (unsigned char*)ws, descs_dev, n, total);
  return hipGetLastError();
}

size_t vy_split_weight_bytes(int cout, int taps, int cin) { return (size_t)((cout + 31) / 32) * taps * (cin / 16) * 3072; }

hipError_t vy_launch_split_weights(const float* w, void* img, int cout, int taps, int cin, hipStream_t s) {
  if (cout % 32 != 0 || cin % 16 != 0) return hipErrorInvalidValue;
  const long long total = (long long)cout * taps * cin;
  const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(split_weights_kernel, dim3(blocks), dim3(256), 0, s, w, (unsigned short*)img, cout, taps, cin);
  return hipGetLastError();
}

// NSA = LDS stages of the A tile (written one k-step ahead: two suffice; three where they fit let the loop unroll by 3
// instead of 6); the W tile has three (its DMA is issued two k-steps ahead).
template <int BM, int BN, int NSA>
// ksplit > 1: split-K for launches of few tiles (a single frame's deep layers): block (tile, s) runs k-steps
// [s T / ksplit, (s + 1) T / ksplit) and stores its raw accumulators to slab s of a.splitk_slabs ([ksplit][M][N] fp32);
// splitk_finish_kernel adds the slabs in order and applies the epilogue.  (The exact kernel cannot do this — its fma
// chain is pinned — which is why a single 608x608 frame is bound by the chain of ONE wave per SIMD there.)
__global__ __launch_bounds__(256, 2) void conv_split_kernel(const ConvArgs a, const int tiles_n, const int ksplit) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WM = 2, WN = 2, NW = 4, NT = 256;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int NSW = 3;
  constexpr int A_PL = BM * 32, W_PL = BN * 32;          // bytes of one plane of a stage (rows x 32 B)
  constexpr int A_ST = 3 * A_PL, W_ST = 3 * W_PL;
  constexpr int W_BASE = NSA * A_ST;                     // LDS: [A stages][W stages][row tables]
  constexpr int W_TOTAL = (BN / 32) * 3;                 // LDS-DMA instructions per block per k-step ...
  constexpr int W_INSTR = (W_TOTAL + NW - 1) / NW;       // ... and per wave (the last ones only on some waves)
  constexpr int A_SETS = BM * 2 / NT;                    // (row, octet) pairs per thread per k-step
  static_assert(A_SETS >= 1 && TM >= 1 && TN >= 1 && (NSA == 2 || NSA == 3), "tile");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSA * A_ST + NSW * W_ST + BM * 16];
  long long* in_off = reinterpret_cast<long long*>(smem + NSA * A_ST + NSW * W_ST);
  unsigned* o_off = reinterpret_cast<unsigned*>(in_off + BM);
  unsigned* r_off = o_off + BM;
  constexpr unsigned kInvalidRow = 0x80000000u;

  // XCD-aware order (conv_igemm.hip): blocks L, L+8, ... share an L2; contiguous run of tiles per XCD, n fastest
  int vblk;
  {
    const int nblk = gridDim.x, L = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, idx = L >> 3;
    vblk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, lrow = lane & 31;
  const int cch = a.Kc >> 4;  // 16-channel chunks per tap
  const int n_tiles = (int)(gridDim.x / (unsigned)ksplit);
  const int ks_idx = vblk / n_tiles, vtile = vblk - ks_idx * n_tiles;  // k-slice major: one slice's tiles are neighbours in time
  const int tile_m = vtile / tiles_n, tile_n = vtile - tile_m * tiles_n;
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

  // A side: thread -> (row, channel octet) of the tile; pointer to the row's centre pixel
  const float* a_ptr[A_SETS];
  unsigned a_lds[A_SETS];  // byte offset inside a stage's plane 0
#pragma unroll
  for (int q = 0; q < A_SETS; ++q) {
    const int idx = q * NT + tid, row = idx >> 1, oct = idx & 1;
    a_ptr[q] = a.in + in_off[row] + oct * 8;
    a_lds[q] = (unsigned)(row * 32 + (VY_SPLIT_SLOT(row, oct) << 4));
  }
  // W side: DMA instruction q = j * NW + wave covers row group g = q / 3 of plane p = q % 3
  const int KS = a.w_taps * cch;
  unsigned w_voff[W_INSTR], w_lds[W_INSTR];
#pragma unroll
  for (int j = 0; j < W_INSTR; ++j) {
    const int q = j * NW + wave, g = q / 3, p = q - g * 3;
    w_voff[j] = (unsigned)(g * KS * 3072 + p * 1024 + lane * 16);
    w_lds[j] = (unsigned)(W_BASE + p * W_PL + g * 1024);
  }
  const unsigned char* w_tile = reinterpret_cast<const unsigned char*>(a.w_split) + (long long)(n0 >> 5) * KS * 3072;

  constexpr bool M16 = VY_SPLIT_M16 != 0;
  constexpr int TS = M16 ? 16 : 32;                  // rows / columns of one MFMA tile
  constexpr int TMs = BM / WM / TS, TNs = BN / WN / TS, NR = TS * TS / 64;  // tiles per wave, accumulator registers per tile
  typedef float accv __attribute__((ext_vector_type(NR)));
  accv acc[TMs][TNs];
#pragma unroll
  for (int i = 0; i < TMs; ++i)
#pragma unroll
    for (int j = 0; j < TNs; ++j)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.0f;

  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const int T_all = a.ntaps * cch;
  const int kb = (int)((long long)ks_idx * T_all / ksplit), ke = (int)((long long)(ks_idx + 1) * T_all / ksplit);
  const int T = ke - kb;  // k-steps of this block (>= 1: the launcher keeps ksplit <= T_all)

  // wave-uniform k-step state
  int n_tap = kb / cch, n_cc = kb - (kb / cch) * cch;
  int a_koff = 0;
  long long w_koff = 0;
  auto advance = [&]() {
    const int tdy = (int)((a.pk_dy >> (2 * n_tap)) & 3u) - 1, tdx = (int)((a.pk_dx >> (2 * n_tap)) & 3u) - 1;
    const int tw = (int)((a.pk_w >> (4 * n_tap)) & 15ull);
    a_koff = (tdy * a.a_Wp + tdx) * a.a_cs + n_cc * 16;
    w_koff = (long long)(tw * cch + n_cc) * 3072;
    if (++n_cc == cch) {
      n_cc = 0;
      ++n_tap;
    }
  };
  f32x4 av[A_SETS][2];
  auto load_a = [&]() {
#pragma unroll
    for (int q = 0; q < A_SETS; ++q) {
      const f32x4* p = reinterpret_cast<const f32x4*>(a_ptr[q] + a_koff);
      av[q][0] = p[0];
      av[q][1] = p[1];
    }
  };
  auto dma_w = [&](int stage) {
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j)
      if (W_TOTAL % NW == 0 || j * NW + wave < W_TOTAL)  // (every wait on these is vmcnt(0): the count may differ per wave)
        lds_dma16_s(w_voff[j], reinterpret_cast<const float*>(w_tile + w_koff), lds0 + stage * W_ST + w_lds[j]);
  };
  auto store_a = [&](int stage) {
#pragma unroll
    for (int q = 0; q < A_SETS; ++q) {
      vy_u32x4 H, M, L;
      split8(av[q][0], av[q][1], H, M, L);
      unsigned char* d = smem + stage * A_ST + a_lds[q];
      *reinterpret_cast<vy_u32x4*>(d) = H;
      *reinterpret_cast<vy_u32x4*>(d + A_PL) = M;
      *reinterpret_cast<vy_u32x4*>(d + 2 * A_PL) = L;
    }
  };
  // fragment read offsets (inside a stage): row * 32 + octet slot.  32x32x16: lane (row = lane & 31, octet = lane >> 5),
  // the plane is an immediate.  16x16x32: lane (row = lane & 15, octet = (lane >> 4) & 1), and lanes 32-63 read the
  // SECOND plane of the instruction's pair: A pairs (l|h) (m|m) (h|h), W pairs (h|l) (m|h) — so that
  //   A(l|h) W(h|l) = l h + h l,   A(m|m) W(m|h) = m m + m h,   A(h|h) W(m|h) = h m + h h     (plane 0 = h, 1 = m, 2 = l)
  const int q16 = lane >> 4, r16 = lane & 15, up = lane >> 5;
  const unsigned fa = M16 ? (unsigned)((wm * (BM / WM) + r16) * 32 + ((q16 & 1) << 4))
                          : (unsigned)((wm * (BM / WM) + lrow) * 32 + (VY_SPLIT_SLOT(lrow, h) << 4));
  const unsigned fw = M16 ? (unsigned)(W_BASE + (wn * (BN / WN) + r16) * 32 + ((q16 & 1) << 4))
                          : (unsigned)(W_BASE + (wn * (BN / WN) + lrow) * 32 + (VY_SPLIT_SLOT(lrow, h) << 4));
  const unsigned fa_lh = fa + (up ? 0 : 2) * A_PL, fa_mm = fa + A_PL, fa_hh = fa;
  const unsigned fw_hl = fw + (up ? 2 : 0) * W_PL, fw_mh = fw + (up ? 0 : 1) * W_PL;

  // the matrix work of one k-step on the A / W stages at `sa` / `sw`
  auto compute = [&](const unsigned char* sa, const unsigned char* sw) {
    if constexpr (M16) {
      auto prod16 = [&](unsigned oa, unsigned ow) {
        bf16x8 af[TMs], wf[TNs];
#pragma unroll
        for (int i = 0; i < TMs; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + oa + i * 512);
#pragma unroll
        for (int j = 0; j < TNs; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(sw + ow + j * 512);
#pragma unroll
        for (int i = 0; i < TMs; ++i)
#pragma unroll
          for (int j = 0; j < TNs; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], wf[j], acc[i][j], 0, 0, 0);
      };
      prod16(fa_lh, fw_hl);  // smallest terms first
      prod16(fa_mm, fw_mh);
      prod16(fa_hh, fw_mh);
    } else {
      // hipcc re-issues fragment reads per product group (24 ds_read_b128 per wave and k-step, where 12 would do) and
      // sinks the MFMAs of this k-step below the next barrier, next to the next k-step's split arithmetic.  Pinning the
      // fragments in registers (12 reads, all waited for before the first MFMA) measured 1.5 % slower.
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
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[pa][i], wf[pw][j], acc[i][j], 0, 0, 0);
      };
      // smallest terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); plane 0 = h, 1 = m, 2 = l
      rd_a(2);
      rd_w(0);
      rd_a(0);
      rd_w(2);
      rd_a(1);
      rd_w(1);
      prod(2, 0);
      prod(0, 2);
#if !defined(VY_SPLIT_DROP_MM)  // (probe builds only: the five-product variant, profiles/r04_negative_results.txt)
      prod(1, 1);
#endif
      prod(1, 0);
      prod(0, 1);
      prod(0, 0);
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using S2 = std::integral_constant<int, 2>;
  using Y = std::true_type;
  using N_ = std::false_type;
  {
    // one k-step on W stage `stw` / A stage `sta` (std::integral_constant in the unrolled main loop, so that the stage
    // offsets fold into the ds_read / ds_write immediates; plain ints in the tail)
    auto kstep = [&](auto stw_, auto sta_, auto has1_, auto has2_) {
      const int STW = stw_, STA = sta_;
      constexpr bool HAS1 = decltype(has1_)::value, HAS2 = decltype(has2_)::value;
      lds_barrier();  // this k-step's stages complete (A by ds_write, W by DMA waited for one k-step ago) and visible
      if (HAS1) {     // k-step t+1: its A values arrived while the previous k-step computed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        store_a((STA + 1) % NSA);
      }
      if (HAS2) {     // k-step t+2: issue its loads now, a whole k-step of matrix work ahead of their use
        advance();
        load_a();
        dma_w((STW + 2) % NSW);
      }
      compute(smem + STA * A_ST, smem + STW * W_ST);
    };
    // prologue: k-step 0 into stage 0 (A through registers, W by DMA), k-step 1's loads in flight
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
    if constexpr (NSA == 3) {
      for (; t + 5 <= T; t += 3) {  // t+2, t+3, t+4 all have a k-step two ahead
        kstep(S0{}, S0{}, Y{}, Y{});
        kstep(S1{}, S1{}, Y{}, Y{});
        kstep(S2{}, S2{}, Y{}, Y{});
      }
    } else {
      for (; t + 8 <= T; t += 6) {
        kstep(S0{}, S0{}, Y{}, Y{});
        kstep(S1{}, S1{}, Y{}, Y{});
        kstep(S2{}, S0{}, Y{}, Y{});
        kstep(S0{}, S1{}, Y{}, Y{});
        kstep(S1{}, S0{}, Y{}, Y{});
        kstep(S2{}, S1{}, Y{}, Y{});
      }
    }
    for (; t < T; ++t) {  // tail (and launches too short for the unrolled loop): stage numbers at run time
      const int after = T - 1 - t, stw = t % NSW, sta = t % NSA;
      if (after >= 2) kstep(stw, sta, Y{}, Y{});
      else if (after == 1) kstep(stw, sta, Y{}, N_{});
      else kstep(stw, sta, N_{}, N_{});
    }
  }

  if (ksplit > 1) {  // raw partial sums to this k-slice's slab; the epilogue belongs to splitk_finish_kernel
    float* slab = a.splitk_slabs + (long long)ks_idx * a.M * a.N;
#pragma unroll
    for (int j = 0; j < TNs; ++j) {
      const int n = n0 + wn * (BN / WN) + j * TS + (M16 ? r16 : lrow);
#pragma unroll
      for (int i = 0; i < TMs; ++i)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int m = m0 + wm * (BM / WM) + i * TS + (M16 ? 4 * q16 + r : (r & 3) + 8 * (r >> 2) + 4 * h);
          if (m < a.M && n < a.N) slab[(long long)m * a.N + n] = acc[i][j][r];
        }
    }
    return;
  }
  // ---- epilogue (conv_igemm.hip): affine -> leaky -> + addend -> store (x1 or x2-replicated)
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
    for (int j = 0; j < TNs; ++j) {
      // C/D maps: 32x32: column lane & 31, row (r & 3) + 8 (r >> 2) + 4 (lane >> 5); 16x16: column lane & 15, row 4 (lane >> 4) + r
      const int ncol = wn * (BN / WN) + j * TS + (M16 ? r16 : lrow);
      const int n = n0 + ncol;
      const bool nvalid = n < a.N;
      const int nc = nvalid ? n : a.N - 1;
      const unsigned colc = (unsigned)ncol * 4u | (nvalid ? 0u : kInvalidRow);
      float sc = 1.0f, sh = 0.0f;
      if (has_scale) sc = a.scale[nc];
      if (has_scale || has_shift) sh = a.shift[nc];
#pragma unroll
      for (int i = 0; i < TMs; ++i) {
        unsigned oo[NR];
        float rv[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const int row = wm * (BM / WM) + i * TS + (M16 ? 4 * q16 + r : (r & 3) + 8 * (r >> 2) + 4 * h);
          oo[r] = __builtin_elementwise_add_sat(o_off[row], colc);
          if (has_res) rv[r] = buf_load_f32(res_rsrc, __builtin_elementwise_add_sat(r_off[row], colc));
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
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
    if (f_scale && f_shift && f_leaky && !f_ups2) {            // conv + BN + leaky (+ residual): inference cells
      if (f_res) epilogue(T_{}, T_{}, T_{}, T_{}, F_{});
      else epilogue(T_{}, T_{}, T_{}, F_{}, F_{});
    } else if (f_scale && f_shift && f_leaky && !f_res) {      // transition cells: x2-replicated store
      epilogue(T_{}, T_{}, T_{}, F_{}, T_{});
    } else if (!f_scale && !f_leaky && !f_ups2) {              // raw conv (training forward) / bias / gradients (+ accumulate)
      if (f_shift) {
        if (f_res) epilogue(F_{}, T_{}, F_{}, T_{}, F_{});
        else epilogue(F_{}, T_{}, F_{}, F_{}, F_{});
      } else {
        if (f_res) epilogue(F_{}, F_{}, F_{}, T_{}, F_{});
        else epilogue(F_{}, F_{}, F_{}, F_{}, F_{});
      }
    } else {
      __builtin_trap();
    }
  }
  // train-mode BatchNorm: per-tile column sums of the raw accumulators in double, as conv_igemm.hip writes them
  // ([tile_m][2][N]; fixed order inside the tile, tiles combined in order by the finalize kernel)
  if constexpr (!M16) {
    if (a.stats) {
      double s1[TNs], s2[TNs];
#pragma unroll
      for (int j = 0; j < TNs; ++j) {
        s1[j] = 0.0;
        s2[j] = 0.0;
#pragma unroll
        for (int i = 0; i < TMs; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const double vv = o_off[row] != kInvalidRow ? (double)acc[i][j][r] : 0.0;
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
        for (int j = 0; j < TNs; ++j) {
          const int col = wn * (BN / WN) + j * 32 + lrow;
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
  }
#endif
}

bool vy_conv_split_supported(const ConvArgs& a) {
  if (!a.w_split) return false;
  if (a.stats && VY_SPLIT_M16) return false;
  if (a.Kc % 32 != 0 || a.N % 64 != 0 || a.ntaps < 1 || a.ntaps > 9) return false;
  // epilogues the kernel instantiates (the exact kernel's set): BN cell (+ residual | x2-replicated), or plain
  // (raw / bias / gradient, + accumulate)
  const bool bn_cell = a.scale && a.shift && a.leaky, plain = !a.scale && !a.leaky;
  return (bn_cell && (a.ups != 2 || !a.res)) || (plain && a.ups != 2);
}

// block tile and k-split of a launch: the cost model's choice (conv_cost_model.h) among 128x128, 128x64 and, for the
// 64-channel layers, 256x64; k-split > 1 only where the slabs fit the scratch the net provides
static long long split_max_ksplit(const ConvArgs& a) {
  if (!a.splitk_slabs || a.stats) return 1;  // (the per-tile statistics need whole sums)
  return std::max<long long>(1, (long long)(a.splitk_bytes / ((unsigned long long)a.M * a.N * 4ull)));
}

void vy_conv_split_cfg(const ConvArgs& a, int* bm, int* bn, int* ksplit) {
  static const char* force = getenv("VY_SPLIT_FORCE");  // experiments: VY_SPLIT_FORCE=128x64 or 128x64x4 (k-split)
  int fks = 1, fbm = 0, fbn = 0;
  if (force && sscanf(force, "%dx%dx%d", &fbm, &fbn, &fks) >= 2 && a.N % fbn == 0 &&
      ((fbm == 128 && (fbn == 128 || fbn == 64)) || (fbm == 256 && fbn == 64))) {  // the three instantiated tiles only
    // a slice is at least 6 k-steps of 16 channels (the kernel's prologue assumes a slice has work: T >= 1)
    const long long t_all = (long long)a.ntaps * (a.Kc >> 4);
    *bm = fbm, *bn = fbn;
    *ksplit = (int)std::max<long long>(1, std::min<long long>(std::min<long long>(std::max(1, fks), split_max_ksplit(a)), t_all / 6));
    return;
  }
  vy_predict_split(a.M, a.N, (double)a.ntaps * a.Kc, (int)std::min<long long>(split_max_ksplit(a), 64), bm, bn, ksplit, vy_args_cus(a));
}

bool vy_conv_split_pays(const ConvArgs& a) {
  if (!vy_conv_split_supported(a)) return false;
  // tests: every supported launch, however small.  Read once per forward by the net (not per launch); per call
  // only for ConvArgs that did not come from a net
  int always_on = a.env_split_always_p1 - 1;
  if (always_on < 0) {
    const char* always = getenv("VY_SPLIT_ALWAYS");
    always_on = always && atoi(always);
  }
  if (always_on) return true;
  const int cus = vy_args_cus(a);
  if (!vy_model_fitted(cus)) return false;  // the comparison between the two kernels was fitted on 256 CUs (conv_cost_model.h)
  int bm, bn, ks;
  return vy_predict_split(a.M, a.N, (double)a.ntaps * a.Kc, (int)std::min<long long>(split_max_ksplit(a), 64), &bm, &bn, &ks, cus) <
         0.97 * vy_conv_predict_us(a);
}

static VyFastDiv split_fastdiv(unsigned d) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  VyFastDiv f;
  f.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 0 ? l - 1 : 0;
  return f;
}

// second launch of a split-K conv: out = epilogue(sum over s of slab s), slabs added in index order (deterministic).
// One thread = 4 consecutive channels of one output pixel.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const ConvArgs a, const int ksplit) {
  const int nq = a.N >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)a.M * nq) return;
  const int m = (int)(idx / nq), n = (int)(idx - (long long)m * nq) << 2;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  const float* p = a.splitk_slabs + (long long)m * a.N + n;
  for (int s = 0; s < ksplit; ++s) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p + (long long)s * a.M * a.N);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] += t[e];
  }
  const int t = (int)fd_div((unsigned)m, a.fd_lw);
  const int x = m - t * a.LW;
  const int b = (int)fd_div((unsigned)t, a.fd_lh);
  const int y = t - b * a.LH;
  const long long pix = (long long)(b * a.o_Hp + y * a.o_s + a.o_oy) * a.o_Wp + x * a.o_s + a.o_ox;
  f32x4 sh = {0.f, 0.f, 0.f, 0.f};
  if (a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + n);
  if (a.scale) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] + sh[e];
  }
  if (a.leaky) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = vy_leaky(v[e]);
  }
  if (a.res) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(a.res + pix * a.r_cs + a.r_co + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] + r[e];
  }
  float* o = a.out + pix * a.o_cs + a.o_co + n;
  *reinterpret_cast<f32x4*>(o) = v;
  if (a.ups == 2) {  // x2-replicated store cropped to the route's size (conv_igemm.hip, row tables)
    const bool dx = 2 * x + 1 < a.o_Wp - 2, dy = 2 * y + 1 < a.o_Hp - 2;
    if (dx) *reinterpret_cast<f32x4*>(o + a.o_cs) = v;
    if (dy) *reinterpret_cast<f32x4*>(o + (long long)a.o_Wp * a.o_cs) = v;
    if (dx && dy) *reinterpret_cast<f32x4*>(o + (long long)(a.o_Wp + 1) * a.o_cs) = v;
  }
}

template <int BM, int BN, int NSA>
static hipError_t launch_split(const ConvArgs& a, int ksplit, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
  hipLaunchKernelGGL((conv_split_kernel<BM, BN, NSA>), dim3(tiles_m * tiles_n * ksplit), dim3(256), 0, s, a, tiles_n, ksplit);
  if (ksplit > 1) {
    const long long work = (long long)a.M * (a.N >> 2);
    hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, a, ksplit);
  }
  return hipGetLastError();
}

hipError_t vy_launch_conv_split(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  if (!vy_conv_split_supported(a) || a.LW < 1 || a.LH < 1 || a.M <= 0) return hipErrorInvalidValue;
  a.fd_lw = split_fastdiv((unsigned)a.LW);
  a.fd_lh = split_fastdiv((unsigned)a.LH);
  a.pk_dy = a.pk_dx = 0;
  a.pk_w = 0;
  for (int t = 0; t < a.ntaps; ++t) {
    if (a.tap_dy[t] < -1 || a.tap_dy[t] > 1 || a.tap_dx[t] < -1 || a.tap_dx[t] > 1 || a.tap_w[t] > 15) return hipErrorInvalidValue;
    a.pk_dy |= (unsigned)(a.tap_dy[t] + 1) << (2 * t);
    a.pk_dx |= (unsigned)(a.tap_dx[t] + 1) << (2 * t);
    a.pk_w |= (unsigned long long)a.tap_w[t] << (4 * t);
  }
  int bm, bn, ks;
  vy_conv_split_cfg(a, &bm, &bn, &ks);
  // (Two more instances were built, measured on the MI355X and removed — profiles/r04_negative_results.txt sections 3 and
  // 7: a deep pipeline for launches that leave a CU with one block, loads three k-steps ahead: no gain; the A operand
  // pre-split into bf16 planes and DMA'd like the weights: 3.5-14 % slower.)
  if (bm == 128 && bn == 128) return launch_split<128, 128, 3>(a, ks, s);
  if (bm == 256 && bn == 64) return launch_split<256, 64, 2>(a, ks, s);
  if (bm == 128 && bn == 64) return launch_split<128, 64, 3>(a, ks, s);
  return hipErrorInvalidValue;
}
