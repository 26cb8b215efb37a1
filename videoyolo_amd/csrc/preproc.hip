// preproc.hip — the step in front of the path: frames as decoded (B,H,W,3) uint8 HWC -> the network's
// (B,3,H,W) fp32 NCHW input, normalised.  Replaces mx.nd.image.to_tensor + mx.nd.image.normalize at
// models/definitions/yolo/transforms.py:331-334 (to_tensor: HWC uint8 -> CHW float32 / 255;
// normalize: (x - mean[c]) / std[c], mean (0.485,0.456,0.406), std (0.229,0.224,0.225)).
// HBM-bound byte work: each thread converts 4 consecutive pixels (12 input bytes, read as three
// 32-bit words; three float4 stores, one per channel plane).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/vyolo.h"
#include "net_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void to_tensor_normalize_kernel(const uint8_t* __restrict__ src,
                                                                  float* __restrict__ dst, long long npix4,
                                                                  long long hw, float m0, float m1, float m2,
                                                                  float s0, float s1, float s2) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;  // group of 4 pixels
  if (q >= npix4) return;
  const long long p = q * 4;          // first pixel (global over B*H*W); H*W % 4 == 0 so a group stays in one image
  const long long b = p / hw, r = p - b * hw;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(src + p * 3);
  const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
  uint8_t px[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    px[i] = (uint8_t)(w0 >> (8 * i));
    px[4 + i] = (uint8_t)(w1 >> (8 * i));
    px[8 + i] = (uint8_t)(w2 >> (8 * i));
  }
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = ((float)px[i * 3 + c] / 255.0f - mean[c]) / stdv[c];
    *reinterpret_cast<f32x4*>(dst + (b * 3 + c) * hw + r) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Resize in front of to_tensor + normalize: timage.imresize(src[i], width, height, interp=9) at
// models/definitions/yolo/transforms.py:325-327 = gluoncv imresize -> mxnet _get_interp_method(9) -> OpenCV
// cv::resize on the uint8 HWC frame [UPSTREAM-RECALLED: neither library is vendored or installable here]:
//   both sides enlarged -> INTER_CUBIC, both shrunk -> INTER_AREA, anything else -> INTER_LINEAR,
// each with OpenCV's 8-bit arithmetic (11-bit fixed-point weights for LINEAR / CUBIC, float pixel-area weights
// for AREA; the CPU checker states every rounding step and is what the tests compare with, bit for bit).
// The resized value is rounded to uint8 exactly where the reference's uint8 NDArray would hold it, then
// (v / 255 - mean) / std goes straight to the NCHW plane: the resized frame is never written to memory.
// One thread per destination pixel (3 channels); tables are recomputed per thread from (dx, dy) — they are a few
// float / double operations, against 12-48 source bytes gathered through L1/L2.
namespace {

struct ResizeArgs {
  const uint8_t* src;   // (B, h, w, 3)
  float* dst;           // (B, 3, H, W)
  int B, h, w, H, W;
  int mode;             // 1 linear, 2 cubic, 3 area (fractional), 4 area with integer factors
  int ix, iy;           // mode 4: the integer factors
  float mean[3], stdv[3];
};

__device__ __forceinline__ void src_coord(int d, double scale, int& s, float& f) {
  f = (float)(((double)d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f -= (float)s;
}

__device__ __forceinline__ int sat_short(float v) {
  const float r = rintf(v);
  return (int)fminf(fmaxf(r, -32768.0f), 32767.0f);
}

// Keys cubic, A = -0.75, OpenCV's interpolateCubic operation order; weights as rounded 11-bit fixed point
__device__ __forceinline__ void cubic_tab(int d, double scale, int ssize, int idx[4], int wq[4]) {
  int s;
  float x;
  src_coord(d, scale, s, x);
  const float A = -0.75f;
  float c[4];
  c[0] = ((A * (x + 1.0f) - 5.0f * A) * (x + 1.0f) + 8.0f * A) * (x + 1.0f) - 4.0f * A;
  c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
  c[2] = ((A + 2.0f) * (1.0f - x) - (A + 3.0f)) * (1.0f - x) * (1.0f - x) + 1.0f;
  c[3] = 1.0f - c[0] - c[1] - c[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int i = s - 1 + k;
    idx[k] = i < 0 ? 0 : (i > ssize - 1 ? ssize - 1 : i);
    wq[k] = sat_short(c[k] * 2048.0f);
  }
}

__device__ __forceinline__ void linear_tab(int d, double scale, int ssize, int idx[2], int wq[2]) {
  int s;
  float f;
  src_coord(d, scale, s, f);
  if (s < 0) {
    s = 0;
    f = 0.0f;
  }
  if (s >= ssize - 1) {
    s = ssize - 1;
    f = 0.0f;
  }
  idx[0] = s;
  idx[1] = s + 1 < ssize ? s + 1 : ssize - 1;
  wq[0] = sat_short((1.0f - f) * 2048.0f);
  wq[1] = sat_short(f * 2048.0f);
}

// computeResizeAreaTab for one destination index: up to kAreaMax (source index, weight) entries in OpenCV's order
constexpr int kAreaMax = 12;  // scale factors up to 10 (a 4096-wide frame to 416)
__device__ __forceinline__ int area_tab(int d, double scale, int ssize, int si[kAreaMax], float al[kAreaMax]) {
  const double f1 = (double)d * scale, f2 = f1 + scale;
  const double cell = fmin(scale, (double)ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = s2 < ssize - 1 ? s2 : ssize - 1;
  s1 = s1 < s2 ? s1 : s2;
  int n = 0;
  if ((double)s1 - f1 > 1e-3 && n < kAreaMax) {
    si[n] = s1 - 1;
    al[n++] = (float)(((double)s1 - f1) / cell);
  }
  for (int sx = s1; sx < s2 && n < kAreaMax; ++sx) {
    si[n] = sx;
    al[n++] = (float)(1.0 / cell);
  }
  if (f2 - (double)s2 > 1e-3 && n < kAreaMax) {
    si[n] = s2;
    al[n++] = (float)(fmin(fmin(f2 - (double)s2, 1.0), cell) / cell);
  }
  return n;
}

__global__ __launch_bounds__(256) void resize_normalize_kernel(const ResizeArgs a) {
  const int dx = blockIdx.x * 256 + threadIdx.x;
  if (dx >= a.W) return;
  const int dy = blockIdx.y, b = blockIdx.z;
  const uint8_t* S = a.src + (long long)b * a.h * a.w * 3;
  const double sx = (double)a.w / (double)a.W, sy = (double)a.h / (double)a.H;
  int out[3];
  if (a.mode == 1) {
    int xi[2], xw[2], yi[2], yw[2];
    linear_tab(dx, sx, a.w, xi, xw);
    linear_tab(dy, sy, a.h, yi, yw);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int r0 = (int)S[((long long)yi[0] * a.w + xi[0]) * 3 + c] * xw[0] + (int)S[((long long)yi[0] * a.w + xi[1]) * 3 + c] * xw[1];
      const int r1 = (int)S[((long long)yi[1] * a.w + xi[0]) * 3 + c] * xw[0] + (int)S[((long long)yi[1] * a.w + xi[1]) * 3 + c] * xw[1];
      out[c] = ((((yw[0] * (r0 >> 4)) >> 16) + ((yw[1] * (r1 >> 4)) >> 16) + 2) >> 2) & 255;
    }
  } else if (a.mode == 2) {
    int xi[4], xw[4], yi[4], yw[4];
    cubic_tab(dx, sx, a.w, xi, xw);
    cubic_tab(dy, sy, a.h, yi, yw);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      long long val = 0;
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
        long long row = 0;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) row += (long long)S[((long long)yi[ky] * a.w + xi[kx]) * 3 + c] * xw[kx];
        val += row * yw[ky];
      }
      const long long v = (val + (1LL << 21)) >> 22;
      out[c] = (int)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
  } else if (a.mode == 4) {
    int sum[3] = {0, 0, 0};
    for (int yy = 0; yy < a.iy; ++yy)
      for (int xx = 0; xx < a.ix; ++xx) {
        const uint8_t* p = S + ((long long)(dy * a.iy + yy) * a.w + dx * a.ix + xx) * 3;
        sum[0] += p[0];
        sum[1] += p[1];
        sum[2] += p[2];
      }
    if (a.ix == 2 && a.iy == 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) out[c] = (sum[c] + 2) >> 2;
    } else {
      const float scale = 1.0f / (float)(a.ix * a.iy);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = rintf((float)sum[c] * scale);
        out[c] = (int)fminf(fmaxf(v, 0.0f), 255.0f);
      }
    }
  } else {
    int xs[kAreaMax], ys[kAreaMax];
    float xa[kAreaMax], ya[kAreaMax];
    const int nx = area_tab(dx, sx, a.w, xs, xa);
    const int ny = area_tab(dy, sy, a.h, ys, ya);
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int j = 0; j < ny; ++j) {
      float buf[3] = {0.0f, 0.0f, 0.0f};
      for (int k = 0; k < nx; ++k) {
        const uint8_t* p = S + ((long long)ys[j] * a.w + xs[k]) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) buf[c] = buf[c] + (float)p[c] * xa[k];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] = j == 0 ? ya[j] * buf[c] : acc[c] + ya[j] * buf[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = (int)fminf(fmaxf(rintf(acc[c]), 0.0f), 255.0f);
  }
  const long long hw = (long long)a.H * a.W;
#pragma unroll
  for (int c = 0; c < 3; ++c)
    a.dst[((long long)b * 3 + c) * hw + (long long)dy * a.W + dx] = ((float)out[c] / 255.0f - a.mean[c]) / a.stdv[c];
}

}  // namespace

extern "C" int vy_preprocess_resize_frames(const uint8_t* frames_hwc, int32_t src_height, int32_t src_width,
                                           float* out_nchw, int32_t batch, int32_t height, int32_t width,
                                           const float* mean3, const float* std3, void* stream) {
  if (!frames_hwc || !out_nchw || !mean3 || !std3 || batch < 1 || height < 1 || width < 1 || src_height < 1 ||
      src_width < 1)
    return fail(VY_ERR_INVALID, "bad argument");
  if (src_height == height && src_width == width)  // cv::resize to the same size is a copy
    return vy_preprocess_frames(frames_hwc, out_nchw, batch, height, width, mean3, std3, stream);
  ResizeArgs a;
  a.src = frames_hwc;
  a.dst = out_nchw;
  a.B = batch;
  a.h = src_height;
  a.w = src_width;
  a.H = height;
  a.W = width;
  a.ix = a.iy = 0;
  // mxnet.image._get_interp_method(9, (oh, ow, nh, nw)) in OpenCV's numbering
  if (height > src_height && width > src_width) {
    a.mode = 2;
  } else if (height < src_height && width < src_width) {
    a.mode = 3;
    const double sx = (double)src_width / width, sy = (double)src_height / height;
    const int ix = (int)nearbyint(sx), iy = (int)nearbyint(sy);
    if (fabs(sx - ix) < 2.220446049250313e-16 && fabs(sy - iy) < 2.220446049250313e-16) {
      a.mode = 4;
      a.ix = ix;
      a.iy = iy;
    } else if (sx > kAreaMax - 2 || sy > kAreaMax - 2) {
      return fail(VY_ERR_UNSUPPORTED, "area resize by more than x%d is not supported", kAreaMax - 2);
    }
  } else {
    a.mode = 1;
  }
  for (int c = 0; c < 3; ++c) {
    a.mean[c] = mean3[c];
    a.stdv[c] = std3[c];
  }
  hipLaunchKernelGGL(resize_normalize_kernel, dim3((width + 255) / 256, height, batch), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int vy_preprocess_frames(const uint8_t* frames_hwc, float* out_nchw, int32_t batch, int32_t height,
                                    int32_t width, const float* mean3, const float* std3, void* stream) {
  if (!frames_hwc || !out_nchw || !mean3 || !std3 || batch < 1 || height < 1 || width < 1)
    return fail(VY_ERR_INVALID, "bad argument");
  const long long hw = (long long)height * width;
  if (hw % 4) return fail(VY_ERR_INVALID, "height*width must be a multiple of 4");
  const long long n4 = (long long)batch * hw / 4;
  hipLaunchKernelGGL(to_tensor_normalize_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), frames_hwc, out_nchw, n4, hw, mean3[0], mean3[1], mean3[2],
                     std3[0], std3[1], std3[2]);
  HIP_TRY(hipGetLastError());
  return 0;
}
