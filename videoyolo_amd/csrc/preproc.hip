// preproc.hip — the step in front of the path: frames as decoded (B,H,W,3) uint8 HWC -> the network's
// (B,3,H,W) fp32 NCHW input, normalised.  Replaces mx.nd.image.to_tensor + mx.nd.image.normalize at
// models/definitions/yolo/transforms.py:331-334 (to_tensor: HWC uint8 -> CHW float32 / 255;
// normalize: (x - mean[c]) / std[c], mean (0.485,0.456,0.406), std (0.229,0.224,0.225)).
// HBM-bound byte work: each thread converts 4 consecutive pixels (12 input bytes, read as three
// 32-bit words; three float4 stores, one per channel plane).
#include <hip/hip_runtime.h>
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
