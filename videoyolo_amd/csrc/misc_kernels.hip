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
//   A operand: a block owns `rows` consecutive output rows of one frame; it first copies the 3 x (rows + 2)
//   input rows they touch into LDS (16-B loads, zero halo rows / columns written explicitly), then every tap of
//   every 32-pixel tile is a conflict-free ds_read_b32 — each input value leaves HBM/L2 once per block (round 1
//   read the frame straight from global memory, 27 dword loads per pixel: FETCH_SIZE 15.8 x the input);
//   B operand: this lane's 14 weights, loaded once per wave;
//   output: lane (cout, h) holds 16 pixels of its channel -> every store writes two 128-B pixel vectors.
// RAW (training): the raw conv goes to the z plane and the block's per-channel sum / sum of squares
// (double) to partials[block][2][32]; otherwise folded-BN affine + LeakyReLU into the activation plane.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// rows per block: 4 (the strip's two halo rows are then half of its payload: 1.5x input re-read instead of 2x), 2 or
// 1 for wide frames (the strip must fit in LDS: 3 x (rows + 2) x (W + 8) floats)
static int stem_rows(int W) {
  const int Wp = ((W + 31) & ~31) + 8;
  if (3 * 6 * Wp * 4 <= 64 * 1024) return 4;
  return 3 * 4 * Wp * 4 <= 120 * 1024 ? 2 : 1;
}

int vy_stem_blocks(int B, int H, int W) { return B * ((H + stem_rows(W) - 1) / stem_rows(W)); }

template <bool RAW>
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a, double* __restrict__ partials, const int rows) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) float strip[];  // [3][rows + 2][W + 8]: data at columns 4 .. W+3
  __shared__ double red[2][4][32];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: descriptors stay scalar
  const int lrow = lane & 31, h = lane >> 5;
  constexpr int kRsrcFlags = 0x00020000;
  // strip row pitch: the frame's width rounded up to whole 32-pixel tiles (+ 4 floats of margin either side); any
  // H and W are accepted — rows past H are skipped, pixels past W are computed on whatever the strip holds there
  // and dropped by the store descriptor's range
  const int Wp = ((a.W + 31) & ~31) + 8, srows = rows + 2;
  const int rpb = (a.H + rows - 1) / rows;       // blocks per frame
  const int b = blockIdx.x / rpb, y0 = (blockIdx.x - b * rpb) * rows;
  // ---- stage the input strip: rows y0-1 .. y0+rows of the three channels
  if ((a.W & 3) == 0) {
    const int w4 = a.W >> 2;
    const int total = 3 * srows * w4;
    for (int i = threadIdx.x; i < total; i += 256) {
      const int q = i % w4, rr = (i / w4) % srows, c = i / (w4 * srows);
      const int y = y0 - 1 + rr;
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (y >= 0 && y < a.H) v = *reinterpret_cast<const f32x4*>(a.x + (((long long)b * 3 + c) * a.H + y) * a.W + q * 4);
      *reinterpret_cast<f32x4*>(strip + (c * srows + rr) * Wp + 4 + q * 4) = v;
    }
  } else {  // rows of a width that is not a multiple of 4 are not 16-byte aligned: element loads
    const int total = 3 * srows * a.W;
    for (int i = threadIdx.x; i < total; i += 256) {
      const int q = i % a.W, rr = (i / a.W) % srows, c = i / (a.W * srows);
      const int y = y0 - 1 + rr;
      strip[(c * srows + rr) * Wp + 4 + q] = (y >= 0 && y < a.H) ? a.x[(((long long)b * 3 + c) * a.H + y) * a.W + q] : 0.0f;
    }
  }
  {
    for (int i = threadIdx.x; i < 3 * srows * 2; i += 256) {  // the zero columns left and right of every row
      const int side = i & 1, rr = i >> 1;
      strip[rr * Wp + (side ? a.W + 4 : 3)] = 0.0f;
    }
  }
  // B operand and, for every k-step of this half-wave (k = 2s + h), the tap's element offset inside the strip
  // relative to the pixel's position in strip row 1 of channel 0
  float bw[14];
  int tap[14];
#pragma unroll
  for (int s = 0; s < 14; ++s) {
    const int k = 2 * s + h;
    const bool kv = k < 27;
    const int kk = kv ? k : 0;
    const int dy = kk / 9 - 1, dx = (kk / 3) % 3 - 1, c = kk % 3;
    bw[s] = kv ? a.w[lrow * 27 + kk] : 0.0f;  // the padded 28th k multiplies a zero weight
    tap[s] = (c * srows + dy) * Wp + dx;
  }
  float sc = 1.0f, sh = 0.0f;
  if (!RAW) {
    if (a.fold_descs) {  // the fold of the whole net rides in this launch (StemArgs): same operations as bn_fold_kernel
      float* P = a.fold_params;
      const FoldDesc d0 = a.fold_descs[a.fold_stem];
      sc = vy_bn_scale(P[d0.gamma + lrow], P[d0.var + lrow], a.fold_eps);
      sh = vy_bn_shift(P[d0.beta + lrow], P[d0.mean + lrow], sc);
      if ((int)blockIdx.x < a.fold_n) {
        const FoldDesc d = a.fold_descs[blockIdx.x];
        for (int c = threadIdx.x; c < d.C; c += 256) {
          const float fs = vy_bn_scale(P[d.gamma + c], P[d.var + c], a.fold_eps);
          P[d.scale + c] = fs;
          P[d.shift + c] = vy_bn_shift(P[d.beta + c], P[d.mean + c], fs);
        }
      }
    } else {
      sc = a.scale[lrow];
      sh = a.shift[lrow];
    }
  }
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;
  const int tiles_per_row = (a.W + 31) / 32;
  const int tiles = rows * tiles_per_row;
  const unsigned st_off = (unsigned)(lrow * 4 + h * 4 * 128);  // this lane's channel, pixel rows 4h.. of the tile
  for (int tile = wave; tile < tiles; tile += 4) {  // wave-uniform
    const int ry = tile / tiles_per_row, x0 = (tile - ry * tiles_per_row) * 32;
    const int y = y0 + ry;
    if (y >= a.H) continue;  // wave-uniform
    const float* centre = strip + (ry + 1) * Wp + 4 + x0 + lrow;
    float av[14];
#pragma unroll
    for (int s = 0; s < 14; ++s) av[s] = centre[tap[s]];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 14; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bw[s], acc, 0, 0, 0);
    // the tile's 32 output pixels are 4 KiB of contiguous memory (32 channels x 4 B per pixel)
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        a.out + ((long long)(b * (a.H + 2) + y + 1) * (a.W + 2) + x0 + 1) * 32, 0,
        (a.W - x0 < 32 ? a.W - x0 : 32) * 128, kRsrcFlags);
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
  if (a.Cout != 32 || a.out_cs != 32 || a.out_co != 0 || a.W < 1 || a.H < 1) return hipErrorInvalidValue;
  return hipSuccess;
}

// rows per block of the inference launch
static int stem_rows_infer(int B, int H, int W) {
  int rows = stem_rows(W);
  const int want = 2 * vy_cu_count();
  while (rows > 1 && B * ((H + rows - 1) / rows) < want) rows >>= 1;
  return rows;
}

bool vy_stem_can_fold(int B, int H, int W, int n_layers) {
  const int rows = stem_rows_infer(B, H, W);
  return B * ((H + rows - 1) / rows) >= n_layers;
}

hipError_t vy_launch_stem(const StemArgs& a, hipStream_t s) {
  if (stem_check(a) != hipSuccess) return hipErrorInvalidValue;
  // A few frames: four rows per block leave most CUs without a block (one 608x608 frame: 152 blocks, 30 us for 47 MB);
  // fewer rows per block until the launch has two blocks per CU (one frame: 608 blocks of one row, the input rows are
  // then staged three times — 13 MB).  Every pixel's fma chain is the same whatever the strip height.  (The training
  // launch below keeps the rule by width alone: its rows are also the layout of the statistics partials.)
  const int rows = stem_rows_infer(a.B, a.H, a.W);
  if (a.fold_descs && (!a.fold_params || a.fold_stem < 0 || a.fold_stem >= a.fold_n || !vy_stem_can_fold(a.B, a.H, a.W, a.fold_n)))
    return hipErrorInvalidValue;
  const size_t lds = (size_t)3 * (rows + 2) * (((a.W + 31) & ~31) + 8) * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(stem_kernel<false>, dim3(a.B * ((a.H + rows - 1) / rows)), dim3(256), lds, s, a, nullptr, rows);
  return hipGetLastError();
}

// training: raw conv -> z plane + partials[vy_stem_blocks][2][32] (double)
hipError_t vy_launch_stem_raw(const StemArgs& a, double* partials, hipStream_t s) {
  if (stem_check(a) != hipSuccess) return hipErrorInvalidValue;
  // RAW mode folds every computed pixel into the BatchNorm partial sums: a partial last 32-pixel tile (whose
  // columns past W come from uninitialised LDS and are only dropped by the store descriptor) would poison them
  if (a.W % 32 != 0) return hipErrorInvalidValue;
  const int rows = stem_rows(a.W);
  const size_t lds = (size_t)3 * (rows + 2) * (((a.W + 31) & ~31) + 8) * sizeof(float);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(stem_kernel<true>, dim3(vy_stem_blocks(a.B, a.H, a.W)), dim3(256), lds, s, a, partials, rows);
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

// dense NCHW -> plane view (vy_net_detect_heads: caller-supplied prediction tensors into the head planes)
__global__ void nchw_to_plane_kernel(const float* src, int B, int H, int W, int cs, int co, int C, float* plane) {
  const long long n = (long long)B * C * H * W;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W);
    long long t = i / W;
    const int y = (int)(t % H);
    t /= H;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    plane[((long long)(b * (H + 2) + y + 1) * (W + 2) + x + 1) * cs + co + c] = src[i];
  }
}

hipError_t vy_launch_nchw_to_plane(const float* src, int B, int H, int W, int cs, int co, int C, float* plane,
                                   hipStream_t s) {
  const long long n = (long long)B * C * H * W;
  long long blocks = (n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(nchw_to_plane_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, B, H, W, cs, co, C, plane);
  return hipGetLastError();
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
