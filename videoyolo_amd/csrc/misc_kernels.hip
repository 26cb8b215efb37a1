// misc_kernels.hip — the small HBM-bound kernels around the conv stack:
//   stem conv (3 -> 32, K = 27: too thin for MFMA tiles; direct conv reading the caller's NCHW image),
//   eval-mode BatchNorm folding, and plane -> NCHW copies for the parity taps.
#include "kernels.h"
#include "../../include/vy_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// Stem: Darknet3D.features[0] = _conv2d(32, 3, 1, 1)  (darknet/three_darknet.py:163-164 over
// layers.py:63-70).  One thread = one output pixel x 32 output channels; the 27-term fma chain per
// channel runs in k = (kh, kw, cin) order like every other conv here.  Reads the NCHW image
// (coalesced along x), writes one 128-B NHWC pixel per thread.
// ---------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
  __shared__ float sw[COUT * 27];
  __shared__ float ssc[COUT], ssh[COUT];
  // output staging: thread t owns row t (its pixel's COUT channels); rows padded to 36 floats so the
  // float4 row writes and the transposed float4 reads below stay 16-B aligned and spread over banks
  __shared__ __attribute__((aligned(16))) float stage[256][COUT + 4];
  __shared__ long long spix[256];
  for (int i = threadIdx.x; i < COUT * 27; i += 256) sw[i] = a.w[i];
  for (int i = threadIdx.x; i < COUT; i += 256) {
    ssc[i] = a.scale[i];
    ssh[i] = a.shift[i];
  }
  __syncthreads();
  const long long npix = (long long)a.B * a.H * a.W;
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p < npix) {
    const int x = (int)(p % a.W);
    const long long t = p / a.W;
    const int y = (int)(t % a.H);
    const int b = (int)(t / a.H);
    float in[27];
    const float* xb = a.x + (long long)b * 3 * a.H * a.W;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iy = y + kh - 1, ix = x + kw - 1;
        const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
        for (int c = 0; c < 3; ++c)
          in[(kh * 3 + kw) * 3 + c] = ok ? xb[((long long)c * a.H + iy) * a.W + ix] : 0.0f;
      }
    spix[threadIdx.x] = ((long long)(b * (a.H + 2) + y + 1) * (a.W + 2) + x + 1) * a.out_cs + a.out_co;
#pragma unroll
    for (int o4 = 0; o4 < COUT; o4 += 4) {
      f32x4 r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc = fmaf(in[k], sw[(o4 + q) * 27 + k], acc);
        acc = fmaf(acc, ssc[o4 + q], ssh[o4 + q]);
        r[q] = vy_leaky(acc);
      }
      *reinterpret_cast<f32x4*>(&stage[threadIdx.x][o4]) = r;
    }
  } else {
    spix[threadIdx.x] = -1;
  }
  __syncthreads();
  // coalesced write-out: 8 consecutive lanes cover one pixel's 128-B channel vector, so every store
  // instruction writes whole 128-B lines (a thread-per-pixel store would touch 64 lines, 16 B each)
  constexpr int CH4 = COUT / 4;
#pragma unroll
  for (int it = 0; it < CH4; ++it) {
    const int lin = it * 256 + threadIdx.x;
    const int pix = lin / CH4, c4 = (lin % CH4) * 4;
    const long long o = spix[pix];
    if (o >= 0) *reinterpret_cast<f32x4*>(a.out + o + c4) = *reinterpret_cast<const f32x4*>(&stage[pix][c4]);
  }
}

hipError_t vy_launch_stem(const StemArgs& a, hipStream_t s) {
  if (a.Cout != 32 || (a.out_cs & 3) || (a.out_co & 3)) return hipErrorInvalidValue;
  const long long npix = (long long)a.B * a.H * a.W;
  hipLaunchKernelGGL(stem_kernel<32>, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, a);
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
