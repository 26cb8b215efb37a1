// misc_kernels.hip — the small HBM-bound kernels around the conv stack:
//   stem conv (3 -> 32, K = 27: one k-step of matrix-core work straight from the caller's NCHW image),
//   eval-mode BatchNorm folding, and plane -> NCHW copies for the parity taps.
#include "kernels.h"
#include "../../include/vy_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Stem: Darknet3D.features[0] = _conv2d(32, 3, 1, 1)  (darknet/three_darknet.py:163-164 over
// layers.py:63-70): 3 -> 32 channels, K = 27.  Too thin for the LDS-tiled kernel (one k-step), but still
// matrix-core work: D[i = pixel][j = cout] = sum_k patch[i][k] * w[j][k] with k = (kh, kw, cin) padded
// 27 -> 28, 14 x v_mfma_f32_32x32x2_f32 per 32 pixels.  The chain runs k ascending (k0 = 2s from lanes
// 0-31, k1 = 2s+1 from lanes 32-63): the same fma order as the CPU checker's 3-channel conv.
//   A operand: lane (pixel, h) reads the NCHW frame directly — 32 consecutive x of one image row
//   (W % 32 == 0, so a 32-pixel tile never straddles rows): coalesced, L1/L2-resident re-reads;
//   B operand: this lane's 14 weights, loaded once per wave;
//   output: lane (cout, h) holds 16 pixels of its channel -> every store writes two 128-B pixel vectors.
// RAW (training): the raw conv goes to the z plane and the block's per-channel sum / sum of squares
// (double) to partials[block][2][32]; otherwise folded-BN affine + LeakyReLU into the activation plane.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kStemTilesPerWave = 8;  // 32-pixel tiles per wave; block = 4 waves = 1024 pixels

int vy_stem_blocks(int B, int H, int W) {
  const long long tiles = (long long)B * H * W / 32;
  return (int)((tiles + 4 * kStemTilesPerWave - 1) / (4 * kStemTilesPerWave));
}

template <bool RAW>
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a, double* __restrict__ partials) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ double red[2][4][32];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: descriptors stay scalar
  const int lrow = lane & 31, h = lane >> 5;
  constexpr unsigned kOob = 0x80000000u;  // beyond every descriptor's range: load returns 0, store is dropped
  constexpr int kRsrcFlags = 0x00020000;
  // B operand and, for every k-step of this half-wave (k = 2s + h), the tap's element offset inside one
  // frame relative to the centre pixel, plus bit masks of the steps whose tap looks up / down / left /
  // right (they read zero padding on the matching image border)
  float bw[14];
  int tap4[14];
  unsigned m_up = 0, m_down = 0, m_left = 0, m_right = 0, m_pad = 0;
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + h;
    const bool kv = k < 27;
    const int kk = kv ? k : 0;
    const int dy = kk / 9 - 1, dx = (kk / 3) % 3 - 1, c = kk % 3;
    bw[s] = kv ? a.w[lrow * 27 + kk] : 0.0f;
    tap4[s] = ((c * a.H + dy) * a.W + dx) * 4;
    m_up |= (unsigned)(dy < 0) << s;
    m_down |= (unsigned)(dy > 0) << s;
    m_left |= (unsigned)(dx < 0) << s;
    m_right |= (unsigned)(dx > 0) << s;
    m_pad |= (unsigned)(!kv) << s;
  }
  float sc = 1.0f, sh = 0.0f;
  if (!RAW) {
    sc = a.scale[lrow];
    sh = a.shift[lrow];
  }
  double s1 = 0.0, s2 = 0.0;
  const long long tiles = (long long)a.B * a.H * a.W / 32;
  const int tiles_per_row = a.W / 32;
  const long long t0 = ((long long)blockIdx.x * 4 + wave) * kStemTilesPerWave;
  const unsigned st_off = (unsigned)(lrow * 4 + h * 4 * 128);  // this lane's channel, pixel rows 4h.. of the tile
  for (int q = 0; q < kStemTilesPerWave; ++q) {
    const long long tile = t0 + q;
    if (tile >= tiles) break;  // wave-uniform
    const int x0 = (int)(tile % tiles_per_row) * 32;
    const long long row = tile / tiles_per_row;
    const int y = (int)(row % a.H), b = (int)(row / a.H);
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + (long long)b * 3 * a.H * a.W, 0, 3 * a.H * a.W * 4, kRsrcFlags);
    const int x = x0 + lrow;
    const unsigned bad = m_pad | (y == 0 ? m_up : 0u) | (y == a.H - 1 ? m_down : 0u) | (x == 0 ? m_left : 0u) |
                         (x == a.W - 1 ? m_right : 0u);
    const int centre4 = (y * a.W + x) * 4;
    float av[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) {
      const unsigned off = ((bad >> s) & 1u) ? kOob : (unsigned)(centre4 + tap4[s]);
      av[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off, 0, 0));
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bw[s], acc, 0, 0, 0);
    // the tile's 32 output pixels are 4 KiB of contiguous memory (32 channels x 4 B per pixel)
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out + ((long long)(b * (a.H + 2) + y + 1) * (a.W + 2) + x0 + 1) * 32, 0, 32 * 128, kRsrcFlags);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r];
      if (RAW) {
        s1 += (double)v;
        s2 += (double)v * (double)v;
      } else {
        v = vy_leaky(fmaf(v, sc, sh));
      }
      // pixel (r&3) + 8*(r>>2) + 4*h of the tile, channel lrow
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), out_rsrc,
                                            st_off + (unsigned)(((r & 3) + 8 * (r >> 2)) * 128), 0, 0);
    }
  }
  if (RAW) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (h == 0) {
      red[0][wave][lrow] = s1;
      red[1][wave][lrow] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int which = threadIdx.x >> 5, c = threadIdx.x & 31;
      partials[(long long)blockIdx.x * 64 + which * 32 + c] =
          (red[which][0][c] + red[which][1][c]) + (red[which][2][c] + red[which][3][c]);
    }
  }
#endif
}

static hipError_t stem_check(const StemArgs& a) {
  // the stem writes a 32-channel plane of its own (activation plane or, in training, the z plane)
  if (a.Cout != 32 || a.out_cs != 32 || a.out_co != 0 || a.W % 32 != 0) return hipErrorInvalidValue;
  return hipSuccess;
}

hipError_t vy_launch_stem(const StemArgs& a, hipStream_t s) {
  if (stem_check(a) != hipSuccess) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stem_kernel<false>, dim3(vy_stem_blocks(a.B, a.H, a.W)), dim3(256), 0, s, a, nullptr);
  return hipGetLastError();
}

// training: raw conv -> z plane + partials[vy_stem_blocks][2][32] (double)
hipError_t vy_launch_stem_raw(const StemArgs& a, double* partials, hipStream_t s) {
  if (stem_check(a) != hipSuccess) return hipErrorInvalidValue;
  hipLaunchKernelGGL(stem_kernel<true>, dim3(vy_stem_blocks(a.B, a.H, a.W)), dim3(256), 0, s, a, partials);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// BatchNorm (eval) -> (scale, shift).  reference: norm_layer(epsilon=1e-5, momentum=0.9) at
// layers.py:68 evaluated outside autograd.train_mode.  One block per BN layer.
// ---------------------------------------------------------------------------------------------
__global__ void bn_fold_kernel(float* params, const FoldDesc* descs, float eps) {
  const FoldDesc d = descs[blockIdx.x];
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const float sc = vy_bn_scale(params[d.gamma + c], params[d.var + c], eps);
    params[d.scale + c] = sc;
    params[d.shift + c] = vy_bn_shift(params[d.beta + c], params[d.mean + c], sc);
  }
}

hipError_t vy_launch_bn_fold(float* params, const FoldDesc* descs_dev, int n_layers, int max_c,
                             float eps, hipStream_t s) {
  (void)max_c;
  hipLaunchKernelGGL(bn_fold_kernel, dim3(n_layers), dim3(256), 0, s, params, descs_dev, eps);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// plane view -> dense NCHW (debug / parity taps only)
// ---------------------------------------------------------------------------------------------
__global__ void plane_to_nchw_kernel(const float* plane, int B, int H, int W, int cs, int co, int C,
                                     float* dst) {
  const long long n = (long long)B * C * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long long t = i / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    dst[i] = plane[((long long)(b * (H + 2) + y + 1) * (W + 2) + x + 1) * cs + co + c];
  }
}

hipError_t vy_launch_plane_to_nchw(const float* plane, int B, int H, int W, int cs, int co, int C,
                                   float* dst, hipStream_t s) {
  const long long n = (long long)B * C * H * W;
  long long blocks = (n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(plane_to_nchw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, plane, B, H, W, cs, co,
                     C, dst);
  return hipGetLastError();
}
