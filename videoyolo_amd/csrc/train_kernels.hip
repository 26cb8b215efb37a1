// train_kernels.hip — the HBM-bound kernels of the training step: BatchNorm (batch statistics,
// forward apply, backward), prediction-bias gradient, stem forward/weight-gradient, the fused
// target-merge + YOLOv3 loss + prediction gradient, and the SGD update.
//
// What they replace in the reference (paths relative to /root/reference):
//   BatchNorm train fwd/bwd   mxnet BatchNorm behind layers.py:68 under autograd.record()
//   dynamic targets + merge   models/definitions/yolo/yolo_target.py:173-205, 226-281
//   loss                      gluoncv.loss.YOLOV3Loss, yolo3.py:994,1187
//   SGD                       gluon.Trainer('sgd', wd, momentum).step(batch_size), train_yolov3.py:527-530,634
// All reductions are two-stage (per-block partials written in a fixed layout, then summed in index
// order in double) so a training step is bit-reproducible run to run; no float atomics.
#include <cstdlib>
#include "kernels.h"
#include "../../include/vy_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const T* __restrict__ partials, int n_part,
                                                               int n_cols, double* __restrict__ out) {
  // block = 32 columns x 32 row-groups; row-group r sums partials r, r+32, ... in order, then the 32
  // group sums are added in order: a fixed summation tree (deterministic)
  __shared__ double red[32][33];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  double acc = 0.0;
  if (c < n_cols)
    for (int t = ry; t < n_part; t += 32) acc += (double)partials[(long long)t * n_cols + c];
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && c < n_cols) {
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += red[r][cx];
    out[c] = s;
  }
}

hipError_t vy_launch_reduce_partials(const float* partials, int n_part, int n_cols, double* out, hipStream_t s) {
  hipLaunchKernelGGL(reduce_partials_kernel<float>, dim3((n_cols + 31) / 32), dim3(1024), 0, s, partials, n_part,
                     n_cols, out);
  return hipGetLastError();
}

hipError_t vy_launch_reduce_partials_f64(const double* partials, int n_part, int n_cols, double* out,
                                         hipStream_t s) {
  hipLaunchKernelGGL(reduce_partials_kernel<double>, dim3((n_cols + 31) / 32), dim3(1024), 0, s, partials, n_part,
                     n_cols, out);
  return hipGetLastError();
}

__global__ void f64_to_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

hipError_t vy_launch_f64_to_f32(const double* src, float* dst, int n, hipStream_t s) {
  hipLaunchKernelGGL(f64_to_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bn_finalize_channel(const BnFinalizeArgs& a, int c, double sum1, double sum2) {
  const double mean = sum1 / a.count;
  double var = sum2 / a.count - mean * mean;  // biased (mxnet BatchNorm)
  if (var < 0.0) var = 0.0;
  const float mf = (float)mean, vf = (float)var;
  const float invstd = 1.0f / sqrtf(vf + a.eps);
  const float sc = a.gamma[c] * invstd;
  a.scale[c] = sc;
  a.shift[c] = fmaf(-mf, sc, a.beta[c]);
  a.save_mean[c] = mf;
  a.save_invstd[c] = invstd;
  a.running_mean[c] = a.running_mean[c] * a.momentum + mf * (1.0f - a.momentum);
  a.running_var[c] = a.running_var[c] * a.momentum + vf * (1.0f - a.momentum);
}

__global__ void bn_finalize_kernel(const BnFinalizeArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  bn_finalize_channel(a, c, a.sums[c], a.sums[a.C + c]);
}

// Per-device BatchNorm (no statistics exchange between the reduce and the finalize): both in one launch.
// Block = kRfCols channels x kRfGroups row groups; row group r sums partial rows r, r + kRfGroups, ... in that
// order (eight loads in flight at a time: the loop is latency-bound, not bandwidth-bound), then the group sums
// are added in group order: a fixed summation tree, for the channel's two columns (sum, sum of squares) at once.
constexpr int kRfCols = 16, kRfGroups = 64;

template <typename T, typename Args, typename Fin>
__device__ __forceinline__ void reduce2_finalize(const T* __restrict__ partials, int n_part, const Args& a, Fin fin) {
  __shared__ double red[2][kRfGroups][kRfCols + 1];
  const int cx = threadIdx.x % kRfCols, ry = threadIdx.x / kRfCols;
  const int c = blockIdx.x * kRfCols + cx;
  double acc1 = 0.0, acc2 = 0.0;
  if (c < a.C) {
    const T* p1 = partials + c;
    const T* p2 = partials + a.C + c;
    const long long rs = 2LL * a.C;
    int t = ry;
    constexpr int U = 8;
    for (; t + (U - 1) * kRfGroups < n_part; t += U * kRfGroups) {
      T v1[U], v2[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v1[u] = p1[(long long)(t + u * kRfGroups) * rs];
        v2[u] = p2[(long long)(t + u * kRfGroups) * rs];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc1 += (double)v1[u];
        acc2 += (double)v2[u];
      }
    }
    for (; t < n_part; t += kRfGroups) {
      acc1 += (double)p1[(long long)t * rs];
      acc2 += (double)p2[(long long)t * rs];
    }
  }
  red[0][ry][cx] = acc1;
  red[1][ry][cx] = acc2;
  __syncthreads();
  if (ry == 0 && c < a.C) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
    for (int r = 0; r < kRfGroups; ++r) {
      s1 += red[0][r][cx];
      s2 += red[1][r][cx];
    }
    fin(a, c, s1, s2);
  }
}

// The BatchNorm passes of the backward pass run on the main stream BESIDE the weight-gradient kernels of the side stream
// (train.hip).  Those hold two MFMA waves per SIMD, and fp32 MFMA shares the vector FMA hardware: a bandwidth-bound pass
// whose waves get the pipe only between 64-cycle matrix instructions crawls (bn_bwd_reduce 17.6 us alone, 150 us beside
// wgrad_kernel<32,128>).  s_setprio raises the issuing priority of these short waves: they take the few vector slots they
// need and the matrix waves lose only those.  VY_BN_PRIO=0 (read once, copied to the device) turns it off for A/B.
__device__ int g_bn_prio = 1;
__device__ __forceinline__ void bn_raise_prio() {
#if defined(__HIP_DEVICE_COMPILE__)
  if (g_bn_prio) __builtin_amdgcn_s_setprio(3);
#endif
}
void vy_bn_prio_init() {
  static bool done = false;
  if (done) return;
  done = true;
  const int v = getenv("VY_BN_PRIO") ? atoi(getenv("VY_BN_PRIO")) : 1;
  if (v != 1) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_bn_prio), &v, sizeof v);
}

__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const double* __restrict__ partials, int n_part,
                                                                  const BnFinalizeArgs a) {
  reduce2_finalize(partials, n_part, a,
                   [](const BnFinalizeArgs& f, int c, double s1, double s2) { bn_finalize_channel(f, c, s1, s2); });
}

// first stage for long partial lists (the 208x208 / 104x104 layers produce thousands of per-tile rows): S row
// slices are summed by S x (n_cols / kRfCols) blocks into out[S][n_cols] (double), which the launch above then
// finishes — two short launches instead of one launch whose C / 16 blocks walk every row.  Order: rows of a
// slice by row group as above, groups in order: fixed.
__global__ __launch_bounds__(1024) void reduce_slices_kernel(const double* __restrict__ partials, int n_part, int n_cols,
                                                             int rows_per_slice, double* __restrict__ out) {
  __shared__ double red[kRfGroups][kRfCols + 1];
  const int cx = threadIdx.x % kRfCols, ry = threadIdx.x / kRfCols;
  const int c = blockIdx.x * kRfCols + cx;
  const int t0 = blockIdx.y * rows_per_slice;
  const int t1 = t0 + rows_per_slice < n_part ? t0 + rows_per_slice : n_part;
  double acc = 0.0;
  if (c < n_cols) {
    int t = t0 + ry;
    constexpr int U = 8;
    for (; t + (U - 1) * kRfGroups < t1; t += U * kRfGroups) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = partials[(long long)(t + u * kRfGroups) * n_cols + c];
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; t < t1; t += kRfGroups) acc += partials[(long long)t * n_cols + c];
  }
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && c < n_cols) {
    double s = 0.0;
#pragma unroll 8
    for (int r = 0; r < kRfGroups; ++r) s += red[r][cx];
    out[(long long)blockIdx.y * n_cols + c] = s;
  }
}

hipError_t vy_launch_bn_reduce_finalize(const double* partials, int n_part, const BnFinalizeArgs& a, double* scratch,
                                        hipStream_t s) {
  // Long per-tile lists go through the slice pass first; up to 2048 rows the finalize kernel's own 64-group reduce is
  // faster than a second launch on the forward chain (threshold 128 -> 2048: forward 10.48 -> 10.30 ms per step)
  static const int min_parts = getenv("VY_REDUCE_MIN") ? atoi(getenv("VY_REDUCE_MIN")) : 2048;
  if (n_part > min_parts && scratch) {
    const int S = VY_REDUCE_SLICES;
    const int rps = (n_part + S - 1) / S;
    const int slices = (n_part + rps - 1) / rps;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((2 * a.C + kRfCols - 1) / kRfCols, slices), dim3(1024), 0, s, partials,
                       n_part, 2 * a.C, rps, scratch);
    partials = scratch;
    n_part = slices;
  }
  hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3((a.C + kRfCols - 1) / kRfCols), dim3(1024), 0, s, partials, n_part,
                     a);
  return hipGetLastError();
}

hipError_t vy_launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((a.C + 127) / 128), dim3(128), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// grid: x over ceil(W*C/4 / 256), y over B*H rows.  One thread = 4 channels of one pixel.
__global__ __launch_bounds__(256) void bn_apply_kernel(const BnApplyArgs a) {
  bn_raise_prio();
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int cq = a.C >> 2;
  if (q >= a.W * cq) return;
  const int x = q / cq, c = (q - x * cq) << 2;
  const int y = blockIdx.y % a.H, b = blockIdx.y / a.H;
  const long long zp = ((long long)(b * (a.H + 2) + y + 1) * (a.W + 2) + x + 1) * a.C + c;
  const f32x4 z = *reinterpret_cast<const f32x4*>(a.z + zp);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + c);
  const long long op = (long long)(b * a.o_Hp + y * a.ups + 1) * a.o_Wp + x * a.ups + 1;
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = vy_leaky(fmaf(z[i], sc[i], sh[i]));
  if (a.res) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(a.res + op * a.r_cs + a.r_co + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = v[i] + r[i];
  }
  float* o = a.out + op * a.o_cs + a.o_co + c;
  *reinterpret_cast<f32x4*>(o) = v;
  if (a.ups == 2) {
    *reinterpret_cast<f32x4*>(o + a.o_cs) = v;
    *reinterpret_cast<f32x4*>(o + (long long)a.o_Wp * a.o_cs) = v;
    *reinterpret_cast<f32x4*>(o + (long long)(a.o_Wp + 1) * a.o_cs) = v;
  }
}

hipError_t vy_launch_bn_apply(const BnApplyArgs& a, hipStream_t s) {
  if ((a.C & 3) || (a.o_cs & 3) || (a.o_co & 3) || (a.r_cs & 3) || (a.r_co & 3)) return hipErrorInvalidValue;
  const int per_row = a.W * (a.C >> 2);
  hipLaunchKernelGGL(bn_apply_kernel, dim3((per_row + 255) / 256, a.B * a.H), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// BN + leaky backward.  Thread layout: CQ = min(C/4, 64) channel quads x PY = 256/CQ pixel lanes.
__device__ __forceinline__ f32x4 load_da(const BnBwdArgs& a, int b, int y, int x, int c) {
  if (a.ups == 1) {
    const long long gp = ((long long)(b * a.g_Hp + y + 1) * a.g_Wp + x + 1) * a.g_cs + a.g_co + c;
    return *reinterpret_cast<const f32x4*>(a.g + gp);
  }
  const long long gp = ((long long)(b * a.g_Hp + 2 * y + 1) * a.g_Wp + 2 * x + 1) * a.g_cs + a.g_co + c;
  const f32x4 g00 = *reinterpret_cast<const f32x4*>(a.g + gp);
  const f32x4 g01 = *reinterpret_cast<const f32x4*>(a.g + gp + a.g_cs);
  const f32x4 g10 = *reinterpret_cast<const f32x4*>(a.g + gp + (long long)a.g_Wp * a.g_cs);
  const f32x4 g11 = *reinterpret_cast<const f32x4*>(a.g + gp + (long long)(a.g_Wp + 1) * a.g_cs);
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (g00[i] + g01[i]) + (g10[i] + g11[i]);
  return r;
}

// Sums of dy and dy*xhat over the pixels, per channel: the two planes (z, and the gradient of the cell's output)
// are read exactly once, 16 B per lane.  Block = CQ channel quads x PY pixel lanes (512 threads); a block owns
// `a.chunk` consecutive image rows (b, y), walks them with incrementally updated coordinates (no division in
// the loop) two pixels at a time (two independent accumulator sets: twice the loads in flight), and writes one
// partial row [2][C]; the ordered second stage (bn_bwd_reduce_finalize) adds the rows in index order.
constexpr int kBwdThreads = 512;

int vy_bn_bwd_rows_per_chunk(int B, int H, int C) {
  const int cq = C >> 2;
  const int CQ = cq < 64 ? cq : 64;
  const int groups = (cq + CQ - 1) / CQ;
  int want = 1024 / groups;  // ~4 blocks per CU over the whole launch, <= 1024 partial rows
  if (want < 1) want = 1;
  const int rows = B * H;
  return (rows + want - 1) / want;
}

int vy_bn_bwd_chunks(const BnBwdArgs& a) { return (a.B * a.H + a.chunk - 1) / a.chunk; }

__global__ __launch_bounds__(kBwdThreads) void bn_bwd_reduce_kernel(const BnBwdArgs a, int CQ) {
  bn_raise_prio();
  __shared__ float red[2][kBwdThreads][4];
  const int PY = kBwdThreads / CQ;
  const int tq = threadIdx.x % CQ, py = threadIdx.x / CQ;
  const int c = (blockIdx.x * CQ + tq) << 2;
  const int rows = a.B * a.H;
  const int r0 = blockIdx.y * a.chunk;
  const int r1 = r0 + a.chunk < rows ? r0 + a.chunk : rows;
  f32x4 s1a = {0, 0, 0, 0}, s2a = {0, 0, 0, 0}, s1b = {0, 0, 0, 0}, s2b = {0, 0, 0, 0};
  if (c < a.C) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(a.save_mean + c);
    const f32x4 is = *reinterpret_cast<const f32x4*>(a.save_invstd + c);
    // pixel lane py starts at pixel py of the chunk and advances by PY pixels = (dq rows, dr columns)
    const int dq = PY / a.W, dr = PY - dq * a.W;
    int row = r0 + py / a.W, x = py % a.W;
    int b = row / a.H, y = row - b * a.H;
    auto advance = [&]() {
      x += dr;
      row += dq;
      y += dq;
      if (x >= a.W) {
        x -= a.W;
        ++row;
        ++y;
      }
      while (y >= a.H) {
        y -= a.H;
        ++b;
      }
    };
    auto accumulate = [&](const f32x4& z, const f32x4& da, f32x4& s1, f32x4& s2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float yv = fmaf(z[i], sc[i], sh[i]);
        const float dy = yv > 0.0f ? da[i] : 0.1f * da[i];
        const float xh = (z[i] - mu[i]) * is[i];
        s1[i] += dy;
        s2[i] = fmaf(dy, xh, s2[i]);
      }
    };
    auto zptr = [&]() {
      return a.z + ((long long)(b * (a.H + 2) + y + 1) * (a.W + 2) + x + 1) * a.C + c;
    };
    while (row < r1) {
      const f32x4 z0 = *reinterpret_cast<const f32x4*>(zptr());
      const f32x4 d0 = load_da(a, b, y, x, c);
      advance();
      if (row < r1) {
        const f32x4 z1 = *reinterpret_cast<const f32x4*>(zptr());
        const f32x4 d1 = load_da(a, b, y, x, c);
        advance();
        accumulate(z0, d0, s1a, s2a);
        accumulate(z1, d1, s1b, s2b);
      } else {
        accumulate(z0, d0, s1a, s2a);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    red[0][threadIdx.x][i] = s1a[i] + s1b[i];
    red[1][threadIdx.x][i] = s2a[i] + s2b[i];
  }
  __syncthreads();
  // fixed tree: pixel lanes py, py+8, py+16, ... first (8 or fewer partial sums per channel quad), then those in order
  const int P8 = PY < 8 ? PY : 8;
  f32x4 u1 = {0, 0, 0, 0}, u2 = {0, 0, 0, 0};
  if (py < P8)
    for (int r = py; r < PY; r += P8)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u1[i] += red[0][r * CQ + tq][i];
        u2[i] += red[1][r * CQ + tq][i];
      }
  __syncthreads();
  if (py < P8) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      red[0][threadIdx.x][i] = u1[i];
      red[1][threadIdx.x][i] = u2[i];
    }
  }
  __syncthreads();
  if (py == 0 && c < a.C) {
    f32x4 t1 = {0, 0, 0, 0}, t2 = {0, 0, 0, 0};
    for (int r = 0; r < P8; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        t1[i] += red[0][r * CQ + tq][i];
        t2[i] += red[1][r * CQ + tq][i];
      }
    float* o = a.partials + (long long)blockIdx.y * 2 * a.C;
    *reinterpret_cast<f32x4*>(o + c) = t1;
    *reinterpret_cast<f32x4*>(o + a.C + c) = t2;
  }
}

hipError_t vy_launch_bn_bwd_reduce(const BnBwdArgs& a, hipStream_t s) {
  if ((a.C & 3) || (a.g_cs & 3) || (a.g_co & 3) || a.chunk < 1) return hipErrorInvalidValue;
  const int cq = a.C >> 2;
  const int CQ = cq < 64 ? cq : 64;
  if (kBwdThreads % CQ) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((cq + CQ - 1) / CQ, vy_bn_bwd_chunks(a)), dim3(kBwdThreads), 0, s, a, CQ);
  return hipGetLastError();
}

__device__ __forceinline__ void bn_bwd_finalize_channel(const BnBwdFinalizeArgs& a, int c, double l1, double l2,
                                                        double g1, double g2) {
  a.dbeta[c] = (float)l1;   // local sums (SyncBN: all-reduced later with the other gradients)
  a.dgamma[c] = (float)l2;
  a.coef[c] = a.gamma[c] * a.save_invstd[c];
  a.coef[a.C + c] = (float)(g1 / a.count);
  a.coef[2 * a.C + c] = (float)(g2 / a.count);
}

__global__ void bn_bwd_finalize_kernel(const BnBwdFinalizeArgs a) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= a.C) return;
  const double* ls = a.local_sums ? a.local_sums : a.sums;
  bn_bwd_finalize_channel(a, c, ls[c], ls[a.C + c], a.sums[c], a.sums[a.C + c]);
}

__global__ __launch_bounds__(1024) void bn_bwd_reduce_finalize_kernel(const float* __restrict__ partials, int n_part,
                                                                      const BnBwdFinalizeArgs a) {
  reduce2_finalize(partials, n_part, a, [](const BnBwdFinalizeArgs& f, int c, double s1, double s2) {
    bn_bwd_finalize_channel(f, c, s1, s2, s1, s2);
  });
}

hipError_t vy_launch_bn_bwd_reduce_finalize(const float* partials, int n_part, const BnBwdFinalizeArgs& a,
                                            hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_reduce_finalize_kernel, dim3((a.C + kRfCols - 1) / kRfCols), dim3(1024), 0, s, partials,
                     n_part, a);
  return hipGetLastError();
}

hipError_t vy_launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((a.C + 127) / 128), dim3(128), 0, s, a);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const BnBwdArgs a) {
  bn_raise_prio();
  const int q = blockIdx.x * 256 + threadIdx.x;
  const int cq = a.C >> 2;
  if (q >= a.W * cq) return;
  const int x = q / cq, c = (q - x * cq) << 2;
  const int y = blockIdx.y % a.H, b = blockIdx.y / a.H;
  float* zp = a.z + ((long long)(b * (a.H + 2) + y + 1) * (a.W + 2) + x + 1) * a.C + c;
  const f32x4 z = *reinterpret_cast<const f32x4*>(zp);
  const f32x4 da = load_da(a, b, y, x, c);
  const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + c);
  const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + c);
  const f32x4 mu = *reinterpret_cast<const f32x4*>(a.save_mean + c);
  const f32x4 is = *reinterpret_cast<const f32x4*>(a.save_invstd + c);
  const f32x4 c1 = *reinterpret_cast<const f32x4*>(a.coef + c);
  const f32x4 c2 = *reinterpret_cast<const f32x4*>(a.coef + a.C + c);
  const f32x4 c3 = *reinterpret_cast<const f32x4*>(a.coef + 2 * a.C + c);
  f32x4 dz;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float yv = fmaf(z[i], sc[i], sh[i]);
    const float dy = yv > 0.0f ? da[i] : 0.1f * da[i];
    const float xh = (z[i] - mu[i]) * is[i];
    dz[i] = c1[i] * ((dy - c2[i]) - xh * c3[i]);
  }
  *reinterpret_cast<f32x4*>(zp) = dz;
}

hipError_t vy_launch_bn_bwd_apply(const BnBwdArgs& a, hipStream_t s) {
  const int per_row = a.W * (a.C >> 2);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((per_row + 255) / 256, a.B * a.H), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// per-channel column sums of a plane view (prediction-conv bias gradient): partials [chunks][C]
int vy_colsum_chunks(int B, int H, int W, int chunk) {
  const long long npix = (long long)B * H * W;
  return (int)((npix + chunk - 1) / chunk);
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ plane, int B, int H, int W, int cs,
                                                     int co, int C, int chunk, float* __restrict__ partials) {
  // threads: 32 channel lanes x 8 pixel lanes; grid.x over channel groups of 32, grid.y over chunks
  __shared__ float red[8][33];
  const int cx = threadIdx.x & 31, py = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cx;
  const long long npix = (long long)B * H * W;
  const long long p0 = (long long)blockIdx.y * chunk;
  long long p1 = p0 + chunk;
  if (p1 > npix) p1 = npix;
  float acc = 0.0f;
  if (c < C)
    for (long long p = p0 + py; p < p1; p += 8) {
      const int x = (int)(p % W);
      const long long t = p / W;
      const int y = (int)(t % H), b = (int)(t / H);
      acc += plane[((long long)(b * (H + 2) + y + 1) * (W + 2) + x + 1) * cs + co + c];
    }
  red[py][cx] = acc;
  __syncthreads();
  if (py == 0 && c < C) {
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][cx];
    partials[(long long)blockIdx.y * C + c] = s;
  }
}

hipError_t vy_launch_colsum(const float* plane, int B, int H, int W, int cs, int co, int C, int chunk,
                            float* partials, hipStream_t s) {
  hipLaunchKernelGGL(colsum_kernel, dim3((C + 31) / 32, vy_colsum_chunks(B, H, W, chunk)), dim3(256), 0, s, plane, B,
                     H, W, cs, co, C, chunk, partials);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// (the stem's training forward, raw conv + per-block channel sums, is stem_kernel<true> in misc_kernels.hip)

// Stem weight gradient: dW[o][k] = sum_p dz[p][o] * patch[p][k], k = (kh,kw,cin) (27) — a 32 x 27 output
// reduced over B*H*W pixels (2.8 M at 416x416 batch 16).  On the matrix core: D[i=o][j=k] += A[i][p] B[p][j],
// two pixels per v_mfma_f32_32x32x2_f32.  Both operands come straight from global memory: lane (o, h) reads
// dz[p+h][o] (a pixel's 32 channels = one 128-B line per half-wave) and lane (k, h) gathers the image value
// of tap k at pixel p+h from the NCHW frame (L1/L2 resident: every input value is used by 27 lanes).
// One wave = kStemWgradPix pixels into one 32x32 accumulator; the four waves of a block are summed through
// LDS in wave order; partials [blocks][32*27] go to the ordered second-stage reduce.
constexpr int kStemWgradPix = 512;
int vy_stem_wgrad_blocks(int B, int H, int W) {
  return (int)(((long long)B * H * W + 4 * kStemWgradPix - 1) / (4 * kStemWgradPix));
}

typedef float f32x16_t __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void stem_wgrad_kernel(const StemWgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ float red[4][32][33];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = lane & 31, h = lane >> 5;
  const long long npix = (long long)a.B * a.H * a.W;
  const long long p0 = ((long long)blockIdx.x * 4 + wave) * kStemWgradPix;
  // this lane's tap (B operand column k = lrow): k = (kh*3 + kw)*3 + c
  const bool kvalid = lrow < 27;
  const int kh = lrow / 9, kw = (lrow / 3) % 3, kc = lrow % 3;
  const int dy = kh - 1, dx = kw - 1;
  // pixel of this lane at step 0: p0 + h; advanced by 2 per step
  long long p = p0 + h;
  int x = (int)(p % a.W);
  long long t = p / a.W;
  int y = (int)(t % a.H), b = (int)(t / a.H);
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  constexpr int U = 8;
  for (int s0 = 0; s0 < kStemWgradPix / 2; s0 += U) {
    float av[U], bv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = p < npix;
      const int iy = y + dy, ix = x + dx;
      const bool inb = ok && kvalid && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      av[u] = ok ? a.dz[((long long)(b * (a.H + 2) + y + 1) * (a.W + 2) + x + 1) * 32 + lrow] : 0.0f;
      bv[u] = inb ? a.x[((long long)(b * 3 + kc) * a.H + iy) * a.W + ix] : 0.0f;
      p += 2;
      x += 2;
      if (x >= a.W) {  // W >= 32: at most one wrap per step
        x -= a.W;
        if (++y == a.H) {
          y = 0;
          ++b;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
  }
  // acc[r] = D[o = (r&3) + 8*(r>>2) + 4*h][k = lrow]
#pragma unroll
  for (int r = 0; r < 16; ++r) red[wave][(r & 3) + 8 * (r >> 2) + 4 * h][lrow] = acc[r];
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 27; e += 256) {
    const int o = e / 27, k = e - o * 27;
    a.partials[(long long)blockIdx.x * 864 + e] = ((red[0][o][k] + red[1][o][k]) + red[2][o][k]) + red[3][o][k];
  }
#endif
}

hipError_t vy_launch_stem_wgrad(const StemWgradArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3(vy_stem_wgrad_blocks(a.B, a.H, a.W)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Fused: decode box (yolo3.py:172-177) -> max IoU with the gt boxes (BBoxBatchIOU, yolo_target.py:202)
// -> dynamic ignore mask (:204) -> merge with the prefetched targets (:263-279) -> YOLOV3Loss terms
// and their derivatives w.r.t. the raw predictions.  One thread per anchor; 256 anchors per block.
constexpr int kLossThreads = 256;
int vy_loss_blocks_per_image(int N) { return (N + kLossThreads - 1) / kLossThreads; }

__device__ __forceinline__ float bce_logits(float x, float z) {
  // relu(x) - x*z + log(1 + exp(-|x|))   (gluon SigmoidBinaryCrossEntropyLoss, from_sigmoid=False)
  const float ax = fabsf(x);
  return (fmaxf(x, 0.0f) - x * z) + vy_logf(1.0f + vy_expf(-ax));
}

__global__ __launch_bounds__(kLossThreads) void loss_kernel(const LossArgs a) {
  extern __shared__ float sgt[];  // M*4 gt boxes of this image
  __shared__ float red[4][kLossThreads / 64];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < a.M * 4; i += kLossThreads) sgt[i] = a.gt_boxes[(long long)b * a.M * 4 + i];
  __syncthreads();
  const int n = blockIdx.x * kLossThreads + threadIdx.x;
  float l_obj = 0.0f, l_ctr = 0.0f, l_scl = 0.0f, l_cls = 0.0f;
  if (n < a.N) {
    int it = n, s = 0;
    for (; s < 3; ++s) {
      const int cnt = a.head[s].H * a.head[s].W * 3;
      if (it < cnt) break;
      it -= cnt;
    }
    const HeadView& hv = a.head[s];
    const int an = it % 3, cell = it / 3;
    const int x = cell % hv.W, y = cell / hv.W;
    const int P = 5 + a.C;
    const long long po = ((long long)(b * (hv.H + 2) + y + 1) * (hv.W + 2) + x + 1) * hv.cs + hv.co + an * P;
    const float* p = hv.pred + po;
    float* dp = a.dpred[s] + po;
    const float rx = p[0], ry = p[1], rw = p[2], rh = p[3], ro = p[4];
    const float sx = vy_sigmoidf(rx), sy = vy_sigmoidf(ry);
    const long long tn = (long long)b * a.N + n;
    const float obj_t = a.obj_t[tn];
    float objness;
    const bool pos = obj_t > 0.0f;
    if (pos) {
      objness = obj_t;
    } else {
      // dynamic target: ignore (-1) when the predicted box overlaps any gt box by more than the threshold
      const float cx = (sx + (float)x) * hv.stride, cy = (sy + (float)y) * hv.stride;
      const float w = vy_expf(rw) * hv.aw[an], h = vy_expf(rh) * hv.ah[an];
      const float hw = w / 2.0f, hh = h / 2.0f;
      const float x1 = cx - hw, y1 = cy - hh, x2 = cx + hw, y2 = cy + hh;
      const float aa = (x2 - x1) * (y2 - y1);
      float best = -1.0f;
      for (int m = 0; m < a.M; ++m) {
        const float gx1 = sgt[m * 4], gy1 = sgt[m * 4 + 1], gx2 = sgt[m * 4 + 2], gy2 = sgt[m * 4 + 3];
        float iw = fminf(x2, gx2) - fmaxf(x1, gx1), ih = fminf(y2, gy2) - fmaxf(y1, gy1);
        iw = iw > 0.0f ? iw : 0.0f;
        ih = ih > 0.0f ? ih : 0.0f;
        const float inter = iw * ih;
        const float ab = (gx2 - gx1) * (gy2 - gy1);
        const float iou = inter / (((aa + ab) - inter) + 1e-15f);
        best = fmaxf(best, iou);
      }
      objness = best > a.ignore_iou_thresh ? -1.0f : 0.0f;
    }
    // objectness
    const float hard = objness > 0.0f ? 1.0f : objness;
    const float omask = objness > 0.0f ? objness : (objness >= 0.0f ? 1.0f : 0.0f);
    l_obj = bce_logits(ro, hard) * omask;
    dp[4] = (vy_sigmoidf(ro) - hard) * omask;
    if (pos) {
      const float ct0 = a.centers_t[tn * 2], ct1 = a.centers_t[tn * 2 + 1];
      const float st0 = a.scales_t[tn * 2], st1 = a.scales_t[tn * 2 + 1];
      const float w0 = a.weights_t[tn * 2] * objness, w1 = a.weights_t[tn * 2 + 1] * objness;
      l_ctr = bce_logits(rx, ct0) * w0 + bce_logits(ry, ct1) * w1;
      dp[0] = (sx - ct0) * w0;
      dp[1] = (sy - ct1) * w1;
      const float d0 = rw - st0, d1 = rh - st1;
      l_scl = fabsf(d0) * w0 + fabsf(d1) * w1;
      dp[2] = (d0 > 0.0f ? w0 : (d0 < 0.0f ? -w0 : 0.0f));
      dp[3] = (d1 > 0.0f ? w1 : (d1 < 0.0f ? -w1 : 0.0f));
      const float smooth = fminf(1.0f / (float)a.C, 1.0f / 40.0f);
      for (int c = 0; c < a.C; ++c) {
        float ct = a.clas_t[tn * a.C + c];
        if (a.label_smooth) {
          if (ct > 0.5f) ct = ct - smooth;
          if (!(ct < -0.5f || ct > 0.5f)) ct = smooth;
        }
        const float cm = (ct >= 0.0f ? 1.0f : 0.0f) * objness;
        const float xc = p[5 + c];
        l_cls += bce_logits(xc, ct) * cm;
        dp[5 + c] = (vy_sigmoidf(xc) - ct) * cm;
      }
    } else {
      dp[0] = 0.0f;
      dp[1] = 0.0f;
      dp[2] = 0.0f;
      dp[3] = 0.0f;
      for (int c = 0; c < a.C; ++c) dp[5 + c] = 0.0f;
    }
  }
  // block reduction (fixed order): wave shuffle, then 4 waves
  float v[4] = {l_obj, l_ctr, l_scl, l_cls};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[q] += __shfl_xor(v[q], off);
    if (lane == 0) red[q][wave] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    const float s = (red[q][0] + red[q][1]) + (red[q][2] + red[q][3]);
    a.partials[((long long)blockIdx.x * a.B + b) * 4 + q] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// Train-mode, non-recording branch (yolo3.py:1189-1192): the per-anchor tensors YOLOOutputV3 returns when
// autograd.is_training() (yolo3.py:162-182) — decoded corner boxes and the RAW centre / scale / objectness /
// class predictions, concatenated over the scales in the order stride 32, 16, 8 -> cell -> anchor.
__global__ __launch_bounds__(256) void raw_preds_kernel(const RawPredArgs a) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= a.N) return;
  int it = n, s = 0;
  for (; s < 3; ++s) {
    const int cnt = a.head[s].H * a.head[s].W * 3;
    if (it < cnt) break;
    it -= cnt;
  }
  const HeadView& hv = a.head[s];
  const int an = it % 3, cell = it / 3;
  const int x = cell % hv.W, y = cell / hv.W;
  const int P = 5 + a.C;
  const float* p = hv.pred + ((long long)(b * (hv.H + 2) + y + 1) * (hv.W + 2) + x + 1) * hv.cs + hv.co + an * P;
  const float rx = p[0], ry = p[1], rw = p[2], rh = p[3];
  const float cx = (vy_sigmoidf(rx) + (float)x) * hv.stride, cy = (vy_sigmoidf(ry) + (float)y) * hv.stride;
  const float hw = vy_expf(rw) * hv.aw[an] / 2.0f, hh = vy_expf(rh) * hv.ah[an] / 2.0f;
  const long long r = (long long)b * a.N + n;
  a.box[r * 4 + 0] = cx - hw;
  a.box[r * 4 + 1] = cy - hh;
  a.box[r * 4 + 2] = cx + hw;
  a.box[r * 4 + 3] = cy + hh;
  a.centers[r * 2 + 0] = rx;
  a.centers[r * 2 + 1] = ry;
  a.scales[r * 2 + 0] = rw;
  a.scales[r * 2 + 1] = rh;
  a.objness[r] = p[4];
  for (int c = 0; c < a.C; ++c) a.class_pred[r * a.C + c] = p[5 + c];
}

hipError_t vy_launch_raw_preds(const RawPredArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(raw_preds_kernel, dim3((a.N + 255) / 256, a.B), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t vy_launch_loss(const LossArgs& a, hipStream_t s) {
  if (a.M < 0 || a.M > 4096) return hipErrorInvalidValue;
  hipLaunchKernelGGL(loss_kernel, dim3(vy_loss_blocks_per_image(a.N), a.B), dim3(kLossThreads),
                     (size_t)(a.M > 0 ? a.M : 1) * 4 * sizeof(float), s, a);
  return hipGetLastError();
}

__global__ void loss_reduce_kernel(const float* __restrict__ partials, int blocks_per_image, int B,
                                   float* __restrict__ losses) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over B*4
  if (i >= B * 4) return;
  const int b = i >> 2, q = i & 3;
  double s = 0.0;
  for (int k = 0; k < blocks_per_image; ++k) s += (double)partials[((long long)k * B + b) * 4 + q];
  losses[(long long)q * B + b] = (float)s;
}

hipError_t vy_launch_loss_reduce(const float* partials, int blocks_per_image, int B, float* losses, hipStream_t s) {
  hipLaunchKernelGGL(loss_reduce_kernel, dim3((B * 4 + 63) / 64), dim3(64), 0, s, partials, blocks_per_image, B,
                     losses);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// SGD with momentum [UPSTREAM-RECALLED mx.optimizer.SGD]: g = rescale*grad + wd*w ;
// mom = momentum*mom - lr*g ; w += mom.
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                                  float* __restrict__ mom, const SgdSeg* __restrict__ segs,
                                                  const int32_t* __restrict__ chunk_seg, float lr, float momentum,
                                                  float wd, float rescale) {
  const SgdSeg sg = segs[chunk_seg[blockIdx.x * 2]];
  if (!sg.enabled) return;
  const long long base = sg.off + (long long)chunk_seg[blockIdx.x * 2 + 1] * VY_SGD_CHUNK;
  const long long end = sg.off + sg.size;
  const float lr_k = lr * sg.lr_mult, wd_k = wd * sg.wd_mult;
  for (int i = threadIdx.x * 4; i < VY_SGD_CHUNK; i += 256 * 4) {
    const long long e = base + i;
    if (e + 3 < end) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(params + e);
      const f32x4 g = *reinterpret_cast<const f32x4*>(grads + e);
      f32x4 m = *reinterpret_cast<const f32x4*>(mom + e);
      f32x4 wn;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float gg = g[q] * rescale + wd_k * w[q];
        m[q] = momentum * m[q] - lr_k * gg;
        wn[q] = w[q] + m[q];
      }
      *reinterpret_cast<f32x4*>(mom + e) = m;
      *reinterpret_cast<f32x4*>(params + e) = wn;
    } else {
      for (int q = 0; q < 4 && e + q < end; ++q) {
        const float gg = grads[e + q] * rescale + wd_k * params[e + q];
        const float mm = momentum * mom[e + q] - lr_k * gg;
        mom[e + q] = mm;
        params[e + q] = params[e + q] + mm;
      }
    }
  }
}

hipError_t vy_launch_sgd(float* params, const float* grads, float* mom, const SgdSeg* segs_dev,
                         const int32_t* chunk_seg_dev, int n_chunks, float lr, float momentum, float wd,
                         float rescale, hipStream_t s) {
  hipLaunchKernelGGL(sgd_kernel, dim3(n_chunks), dim3(256), 0, s, params, grads, mom, segs_dev, chunk_seg_dev, lr,
                     momentum, wd, rescale);
  return hipGetLastError();
}
