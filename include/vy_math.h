/*
 * vy_math.h — exactly-reproducible fp32 scalar math shared by the HIP kernels and by
 * host-side code (BN folding, tests).
 *
 * Every function is a fixed sequence of IEEE-754 binary32 operations (add, mul, fma,
 * correctly rounded divide / sqrt, integer bit moves), so it returns bit-identical
 * results on gfx950 (hipcc -ffp-contract=off, correctly rounded divide/sqrt, f32
 * denormals kept) and on an x86 host (gcc -mfma -ffp-contract=off).  Nothing here
 * calls libm's expf/logf, whose results differ between libraries by an ulp or two;
 * that freedom is what would make "bit-exact NMS indices" a matter of luck.
 *
 * Replaces (numerically, to ~1-2 ulp) the mxnet operators the reference calls at
 *   models/definitions/yolo/yolo3.py:172-175  (F.sigmoid, F.exp in YOLOOutputV3)
 *   models/definitions/layers.py:68-69        (BatchNorm eval affine, LeakyReLU(0.1))
 */
#ifndef VY_MATH_H
#define VY_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define VY_HD __host__ __device__ __forceinline__
#else
#define VY_HD static inline
#endif

VY_HD float vy_bits_to_f32(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
VY_HD uint32_t vy_f32_to_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}

/* exp(x): Cody-Waite reduction by ln2 (two-constant split), degree-6 polynomial in
 * Horner form with explicit fmaf, scale by 2^n through the exponent field.
 * x > 88.5 -> +inf, x < -86 -> 0 (no denormal results; the detector never needs them). */
VY_HD float vy_expf(float x) {
  if (x > 88.5f) return vy_bits_to_f32(0x7f800000u);
  if (!(x >= -86.0f)) return (x != x) ? x : 0.0f;
  const float kLog2e = 1.44269504088896341f;
  const float kLn2Hi = 0.693359375f;
  const float kLn2Lo = -2.12194440e-4f;
  const float kMagic = 12582912.0f; /* 1.5 * 2^23: round-to-nearest-even to integer */
  float t = fmaf(x, kLog2e, kMagic);
  float n = t - kMagic;
  float r = fmaf(n, -kLn2Hi, x);
  r = fmaf(n, -kLn2Lo, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float r2 = r * r;
  p = fmaf(p, r2, r);
  p = p + 1.0f;
  int32_t ni = (int32_t)n;
  if (ni > 127) { /* only x in (88.03, 88.5]: split the scale so the exponent add cannot wrap */
    p = p * 2.0f;
    ni -= 1;
  }
  uint32_t u = vy_f32_to_bits(p) + ((uint32_t)ni << 23);
  return vy_bits_to_f32(u);
}

/* sigmoid(x) = 1 / (1 + exp(-x))   (mxnet's formula; correctly rounded divide) */
VY_HD float vy_sigmoidf(float x) { return 1.0f / (1.0f + vy_expf(-x)); }

/* log(x) for x > 0 (used by the loss: log1p(exp(-|x|)) is evaluated as vy_logf(1 + e)).
 * Reduction x = m * 2^e with m in [sqrt(1/2), sqrt(2)), then log(m) = 2 atanh(s),
 * s = (m-1)/(m+1), odd polynomial in s. */
VY_HD float vy_logf(float x) {
  if (x != x || x < 0.0f) return vy_bits_to_f32(0x7fc00000u);
  if (x == 0.0f) return vy_bits_to_f32(0xff800000u);
  uint32_t u = vy_f32_to_bits(x);
  if (u >= 0x7f800000u) return x;
  int32_t e = 0;
  if (u < 0x00800000u) { /* denormal: scale up by 2^23 */
    x = x * 8388608.0f;
    u = vy_f32_to_bits(x);
    e = -23;
  }
  e += (int32_t)(u >> 23) - 127;
  uint32_t mant = (u & 0x007fffffu) | 0x3f800000u;
  float m = vy_bits_to_f32(mant);
  if (m > 1.41421356f) {
    m = m * 0.5f;
    e += 1;
  }
  float s = (m - 1.0f) / (m + 1.0f);
  float s2 = s * s;
  /* 2*atanh(s)/s = 2 + 2/3 s^2 + 2/5 s^4 + ... ; |s| <= 0.1716 so the s^16 term is < 1e-13 */
  float p = 0.13333333333f;
  p = fmaf(p, s2, 0.15384615385f);
  p = fmaf(p, s2, 0.18181818182f);
  p = fmaf(p, s2, 0.22222222222f);
  p = fmaf(p, s2, 0.28571428571f);
  p = fmaf(p, s2, 0.4f);
  p = fmaf(p, s2, 0.66666666667f);
  p = fmaf(p, s2, 2.0f);
  float lm = p * s;
  float fe = (float)e;
  float r = fmaf(fe, 0.693359375f, lm);
  r = fmaf(fe, -2.12194440e-4f, r);
  return r;
}

/* LeakyReLU(0.1): mxnet LeakyReLU(act_type='leaky', slope=0.1): x > 0 ? x : slope*x */
/* written as max(x, 0.1 x): the same value for every x (x > 0: x > 0.1 x; x < 0: 0.1 x > x; +-0 and NaN map to
 * themselves) in two instructions instead of three on the device epilogue */
VY_HD float vy_leaky(float x) { return fmaxf(x, 0.1f * x); }

/* BatchNorm (eval) folded into one fma: scale = gamma / sqrt(var + eps),
 * shift = beta - mean*scale.  Both sides (host fold, device fold kernel) use these. */
VY_HD float vy_bn_scale(float gamma, float var, float eps) { return gamma / sqrtf(var + eps); }
VY_HD float vy_bn_shift(float beta, float mean, float scale) { return fmaf(-mean, scale, beta); }

/* The pinned summation order of a forward convolution (oracle/ref_ops.c header): its K = taps x Cin
 * products (Cin counted in 32-channel k-steps, so K = taps * 32 * ceil(Cin / 32)) are summed as S independent fp32 fma
 * chains over S contiguous, equally long runs of k-steps, and the chains are then added in run order starting from +0:
 *     out = (((+0 + P0) + P1) + P2) + P3.
 * S depends on the layer's K only — never on the batch, the tile or the launch — so a frame gives the same bits whatever
 * it is batched with:  K >= 4096 -> 4 (the eight 3x3 cells on 512 channels: the longest chains of the net, on its smallest
 * maps), else 1.
 * Why not one chain everywhere (rounds 1-5): a single chain per output element is serial; one frame's 13x13 / 19x19 maps
 * give the chip 48 / 96 tiles of 64x64 for 256 CUs, each a 144-k-step chain — those launches are latency-bound.  Independent
 * runs can go to different workgroups: one 416x416 frame 1.95 -> 1.62 ms, one 608x608 frame 2.23 -> 1.99 ms.
 * Why not more runs, or runs for K >= 2048 too (measured, same box, profiles/r06_runs_ab.txt): a workgroup that computes
 * all runs of a tile itself — every launch at batch 64 — has to park each finished chain in memory (there are no registers
 * for a second accumulator tile) and read it back; all resident workgroups do that at the same moment, so it costs what
 * moving those bytes at HBM speed costs: 0.03 % of the batch-64 step per (launch x parked chain).  4 / 2 runs for
 * K >= 4096 / 2048: one 416 frame 1.50 ms but the batch-64 step -1.1 %; this rule: -0.7 %; two runs for K >= 4096: -0.35 %
 * (one 416 frame 1.75 ms). */
static inline int vy_conv_k_chunks(long long K) { return K >= 4096 ? 4 : 1; }
/* ... for a conv of `taps` taps over `cch` 32-channel k-steps per tap: the runs must be equally long and a multiple of four
 * k-steps (the kernels' LDS pipeline depth); a shape that is not (none of this net's) is summed as one chain. */
static inline int vy_conv_runs(int taps, int cch) {
  const int T = taps * cch, S = vy_conv_k_chunks((long long)T * 32);
  return (S > 1 && T % (4 * S) == 0) ? S : 1;
}

/* IoU of two corner-format boxes, no +1 offset (mxnet box_nms / box_iou, corner format).
 * Returns 0 when the union is not positive (mxnet: u <= 0 ? 0 : i/u). */
VY_HD float vy_box_iou(float ax1, float ay1, float ax2, float ay2, float bx1, float by1, float bx2,
                       float by2) {
  float iw = fminf(ax2, bx2) - fmaxf(ax1, bx1);
  float ih = fminf(ay2, by2) - fmaxf(ay1, by1);
  if (!(iw > 0.0f) || !(ih > 0.0f)) return 0.0f;
  float inter = iw * ih;
  float aa = (ax2 - ax1) * (ay2 - ay1);
  float ab = (bx2 - bx1) * (by2 - by1);
  float uni = (aa + ab) - inter;
  return (uni <= 0.0f) ? 0.0f : inter / uni;
}

#endif /* VY_MATH_H */
