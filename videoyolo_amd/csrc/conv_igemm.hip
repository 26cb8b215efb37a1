// conv_igemm.hip — 3x3 / 1x1 convolution as an implicit GEMM on the gfx950 fp32 matrix core.
//
// Replaces the mxnet operator chain behind the reference's `_conv2d` cell
// (models/definitions/layers.py:63-70: Conv2D(no bias) -> BatchNorm -> LeakyReLU(0.1)), the
// residual add of DarknetBasicBlockV3 (darknet/three_darknet.py:119-123), the 1x1 prediction conv
// with bias (yolo/yolo3.py:62) and `_upsample` + concat (layers.py:11-20, yolo3.py:1167-1177), all
// in ONE kernel: the affine / activation / residual / x2-replicate happen on the accumulator.
//
// GEMM view:  D[m][n] = sum_k A[m][k] * W[n][k]
//   m = (b, oy, ox) output pixel        M = B*Ho*Wo
//   n = output channel                  N = Cout
//   k = (kh, kw, cin)                   K = ks*ks*Cin, Cin % 32 == 0
// A is never materialised: activations are zero-bordered NHWC planes (kernels.h), so the A row of
// pixel m for tap (kh,kw) is 32 contiguous floats at a fixed offset from the pixel's base address.
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain in k order (k0 from lanes 0-31, k1
// from lanes 32-63).  Each lane fetches 4 consecutive k with one ds_read_b128 and the two lane
// halves pick (k0|k1) and (k2|k3), so every output element is ONE fma chain over k ascending —
// the same order the CPU checker uses, which makes the conv stack bit-reproducible.
//
// Tiling (MI355X): 256 threads = 4 waves; block tile 128 x BN x 32(k); per k-step the A and W
// tiles (128 B per row) are brought in by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
// instruction = 8 rows) into a double buffer; the 16-B chunk index is XOR-swizzled on the
// SOURCE side with (row>>1)&7 so the ds_read_b128 of 32 different rows at one k-chunk is
// bank-conflict free.  128x128 tile: 64 KiB of LDS -> 2 blocks / CU, 64 accumulator VGPRs / lane.
#include "kernels.h"
#include "../../include/vy_math.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (amdgcn builtins below)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;  // LDS-DMA instructions per wave per k-step
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TM >= 1 && TN >= 1, "tile");
  // one LDS object: [stage0 A|W][stage1 A|W][row tables]
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + 3 * BM * 8];
  long long* in_off = reinterpret_cast<long long*>(smem + 2 * STAGE);
  long long* out_off = in_off + BM;
  long long* res_off = out_off + BM;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5;

  // XCD-aware tile order: blocks L, L+8, ... share an XCD (L2); give each XCD a contiguous run
  // of tiles with n fastest so neighbours in time re-use the same A rows and the whole W panel.
  int v;
  {
    const int nblk = gridDim.x, L = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, idx = L >> 3;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = v / tiles_n, tile_n = v - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int Hp = a.Hi + 2, Wp = a.Wi + 2;
  const int K = a.ksize * a.ksize * a.Cin;

  if (tid < BM) {
    const int m = m0 + tid;
    const int mm = m < a.M ? m : a.M - 1;
    const int ox = mm % a.Wo;
    const int t = mm / a.Wo;
    const int oy = t % a.Ho;
    const int b = t / a.Ho;
    const int o1 = (a.ksize == 1) ? 1 : 0;
    in_off[tid] = ((long long)(b * Hp + oy * a.stride + o1) * Wp + ox * a.stride + o1) * a.in_cs + a.in_co;
    const int Hop = a.Ho * a.ups + 2, Wop = a.Wo * a.ups + 2;
    out_off[tid] = (m < a.M)
                       ? ((long long)(b * Hop + oy * a.ups + 1) * Wop + ox * a.ups + 1) * a.out_cs + a.out_co
                       : -1;
    res_off[tid] = ((long long)(b * (a.Ho + 2) + oy + 1) * (a.Wo + 2) + ox + 1) * a.res_cs + a.res_co;
  }
  __syncthreads();

  // per-lane LDS-DMA source pointers (tap / k offsets are added per k-step, they are uniform)
  const float* a_src[A_INSTR];
  const float* b_src[B_INSTR];
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_src[j] = a.in + in_off[row] + chunk * 4;
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int row = (j * 4 + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int n = n0 + row;
    n = n < a.Cout ? n : a.Cout - 1;
    b_src[j] = a.w + (long long)n * K + chunk * 4;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int cchunks = a.Cin >> 5;
  const int T = a.ksize * a.ksize * cchunks;

  auto stage = [&](int t, int buf) {
    const int tap = t / cchunks;
    const int cc = t - tap * cchunks;
    const int kh = (a.ksize == 3) ? tap / 3 : 0;
    const int kw = (a.ksize == 3) ? tap - kh * 3 : 0;
    const long long a_koff = (long long)(kh * Wp + kw) * a.in_cs + cc * 32;
    const int b_koff = t * 32;
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j)
      __builtin_amdgcn_global_load_lds(a_src[j] + a_koff, LDS_PTR(sA + (j * 4 + wave) * 1024), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j)
      __builtin_amdgcn_global_load_lds(b_src[j] + b_koff, LDS_PTR(sB + (j * 4 + wave) * 1024), 16, 0, 0);
  };

  stage(0, 0);
  const int lrow = lane & 31;
  for (int t = 0; t < T; ++t) {
    // tile t has landed (own DMA waited, then everyone's via the barrier) and every wave is done
    // reading the other buffer (it computed tile t-1 before arriving here)
    __syncthreads();
    if (t + 1 < T) stage(t + 1, (t + 1) & 1);
    const unsigned char* sA = smem + (t & 1) * STAGE;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float alo[TM], ahi[TM], blo[TN], bhi[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + lrow;
        const f32x4 q = *reinterpret_cast<const f32x4*>(sA + row * 128 + ((kk ^ ((row >> 1) & 7)) << 4));
        alo[i] = h ? q[1] : q[0];
        ahi[i] = h ? q[3] : q[2];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = (wn * TN + j) * 32 + lrow;
        const f32x4 q = *reinterpret_cast<const f32x4*>(sB + row * 128 + ((kk ^ ((row >> 1) & 7)) << 4));
        blo[j] = h ? q[1] : q[0];
        bhi[j] = h ? q[3] : q[2];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(alo[i], blo[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ahi[i], bhi[j], acc[i][j], 0, 0, 0);
    }
  }

  // epilogue: affine (folded BN or bias) -> leaky -> + residual -> store (x1 or x2-replicated)
  const int Wop = a.Wo * a.ups + 2;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + lrow;
    const bool nvalid = n < a.Cout;
    float sc = 1.0f, sh = 0.0f;
    if (nvalid) {
      if (a.scale) sc = a.scale[n];
      if (a.shift) sh = a.shift[n];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long long oo = out_off[row];
        if (oo < 0 || !nvalid) continue;
        float vv = acc[i][j][r];
        if (a.scale)
          vv = fmaf(vv, sc, sh);
        else if (a.shift)
          vv = vv + sh;
        if (a.leaky) vv = vy_leaky(vv);
        if (a.res) vv = vv + a.res[res_off[row] + n];
        float* o = a.out + oo + n;
        o[0] = vv;
        if (a.ups == 2) {
          o[a.out_cs] = vv;
          o[(long long)Wop * a.out_cs] = vv;
          o[(long long)(Wop + 1) * a.out_cs] = vv;
        }
      }
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}


template <int BM, int BN, int WM, int WN>
static hipError_t launch_cfg(const ConvArgs& a, hipStream_t s) {
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.Cout + BN - 1) / BN;
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), dim3(tiles_m * tiles_n), dim3(256), 0, s, a, tiles_n);
  return hipGetLastError();
}

hipError_t vy_launch_conv_igemm(const ConvArgs& a, hipStream_t s) {
  if (a.Cin % 32 != 0 || (a.ksize != 1 && a.ksize != 3) || a.M <= 0) return hipErrorInvalidValue;
  if (a.Cout <= 32) return launch_cfg<128, 32, 4, 1>(a, s);
  if (a.Cout <= 64) return launch_cfg<128, 64, 2, 2>(a, s);
  return launch_cfg<128, 128, 2, 2>(a, s);
}
