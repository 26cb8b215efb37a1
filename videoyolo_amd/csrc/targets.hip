// targets.hip — the prefetch target generator on the device.
//
// Replaces YOLOV3PrefetchTargetGenerator.forward (models/definitions/yolo/yolo_target.py:31-148), which
// the reference runs as a Python double loop in DataLoader workers (transforms.py:259-277): for every
// valid gt box, the best of the 9 anchors by IoU of zero-centred boxes (:86-94) picks the scale and the
// anchor slot, the box centre picks the cell by int() truncation (:115-116), and the five target tensors
// get one row written (:118-133); `_slice` (:139-148) then keeps, per scale, only that scale's three
// anchors, which is the (B, N, .) layout produced directly here (N ordered stride 32, 16, 8 -> cell ->
// anchor, like the network's predictions).
//
// Two launches: a fill (zeros, class targets -1: :79-84) and a scatter with one 64-thread group per
// image.  The reference loop is sequential — a later gt box landing in the same (cell, anchor) slot
// overwrites the earlier one completely, and the loop stops at the first padded row (:107-108); here
// every box computes its slot in parallel and only the LAST valid box of each slot writes.
// Arithmetic types follow the reference under its pinned interpreter (Python 3.6.3, environment.yml:6 =>
// NumPy 1.x scalar promotion): gtx/gty/gtw/gth are np.float32 scalars (fp32 BBoxCornerToCenter), but
// `gtx / orig_width * width` mixes them with Python ints and is evaluated in float64 (:115-119), as is
// `2.0 - gtw * gth / orig_width / orig_height` after the fp32 product (:123) and `1 / anchor` when
// max(gtw, 1) returns the Python int (:121-122).  The cell index is int() of the float64 value and the
// centre target is the float64 remainder rounded to fp32 on the store.  No contraction; fp32 log via
// include/vy_math.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vy_math.h"
#include "../../include/vyolo.h"
#include "net_internal.h"

namespace {

constexpr int kMaxGt = 1024;  // gt rows per image staged in LDS

// anchors in prediction order: stride 32, 16, 8 (wrappers.py:80-84 reversed by yolo3.py:1013-1014)
__constant__ float kAnchorW[9] = {116.f, 156.f, 373.f, 30.f, 62.f, 59.f, 10.f, 16.f, 33.f};
__constant__ float kAnchorH[9] = {90.f, 198.f, 326.f, 61.f, 45.f, 119.f, 13.f, 30.f, 23.f};

struct TargetArgs {
  const float* gt_boxes;  // (B,M,4) corner pixels, rows of -1 = padding
  const float* gt_ids;    // (B,M) class index as float
  const float* gt_mix;    // (B,M) mixup ratio or null
  float *obj, *ctr, *scl, *wts, *cls;
  int B, M, H, W, C, N;
  int fh[3], fw[3], base[3];  // per scale: feature-map size and first row of the scale in N
};

__global__ __launch_bounds__(256) void targets_fill_kernel(const TargetArgs a) {
  const long long rows = (long long)a.B * a.N;
  const long long small = rows * 7;               // obj 1 + ctr 2 + scl 2 + wts 2 floats per row
  const long long total = small + rows * a.C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    if (i < rows)
      a.obj[i] = 0.0f;
    else if (i < rows * 3)
      a.ctr[i - rows] = 0.0f;
    else if (i < rows * 5)
      a.scl[i - rows * 3] = 0.0f;
    else if (i < small)
      a.wts[i - rows * 5] = 0.0f;
    else
      a.cls[i - small] = -1.0f;
  }
}

__global__ __launch_bounds__(64) void targets_scatter_kernel(const TargetArgs a) {
  __shared__ int slot[kMaxGt];
  __shared__ int first_invalid;
  const int b = blockIdx.x, t = threadIdx.x;
  const float* gb = a.gt_boxes + (long long)b * a.M * 4;
  if (t == 0) first_invalid = a.M;
  __syncthreads();
  for (int m = t; m < a.M; m += 64) {
    const float x1 = gb[m * 4 + 0], y1 = gb[m * 4 + 1], x2 = gb[m * 4 + 2], y2 = gb[m * 4 + 3];
    if (!(x1 >= 0.0f && y1 >= 0.0f && x2 >= 0.0f && y2 >= 0.0f)) atomicMin(&first_invalid, m);
  }
  __syncthreads();
  const int nvalid = first_invalid;  // the reference loop breaks at the first invalid row
  for (int m = t; m < nvalid; m += 64) {
    const float x1 = gb[m * 4 + 0], y1 = gb[m * 4 + 1], x2 = gb[m * 4 + 2], y2 = gb[m * 4 + 3];
    const float gw = x2 - x1, gh = y2 - y1;
    const float gx = x1 + gw / 2.0f, gy = y1 + gh / 2.0f;  // gluoncv BBoxCornerToCenter: xmin + width / 2
    // best anchor: first maximum of IoU(zero-centred anchor, zero-centred gt) = argmax (:92-94)
    int match = 0;
    float best = -1.0f;
    for (int k = 0; k < 9; ++k) {
      const float iw = fminf(kAnchorW[k], gw), ih = fminf(kAnchorH[k], gh);
      const float inter = fmaxf(iw, 0.0f) * fmaxf(ih, 0.0f);
      const float ua = (kAnchorW[k] * kAnchorH[k] + gw * gh) - inter;
      const float iou = ua > 0.0f ? inter / ua : 0.0f;
      if (iou > best) {
        best = iou;
        match = k;
      }
    }
    const int l = match / 3;
    const double fx = (double)gx / (double)a.W * (double)a.fw[l];  // float64: np.float32 scalar / Python int
    const double fy = (double)gy / (double)a.H * (double)a.fh[l];
    const int lx = (int)fx, ly = (int)fy;
    const long long cell = (long long)ly * a.fw[l] + lx;
    // a centre on the right / bottom edge indexes past its row: the reference then writes a (cell,
    // anchor) pair that `_slice` discards unless it still lies inside this scale's cell range
    slot[m] = (cell >= 0 && cell < (long long)a.fh[l] * a.fw[l]) ? a.base[l] + (int)cell * 3 + (match - 3 * l) : -1;
  }
  __syncthreads();
  for (int m = t; m < nvalid; m += 64) {
    const int n = slot[m];
    if (n < 0) continue;
    bool last = true;
    for (int q = m + 1; q < nvalid; ++q)
      if (slot[q] == n) {
        last = false;
        break;
      }
    if (!last) continue;
    const float x1 = gb[m * 4 + 0], y1 = gb[m * 4 + 1], x2 = gb[m * 4 + 2], y2 = gb[m * 4 + 3];
    const float gw = x2 - x1, gh = y2 - y1;
    const float gx = x1 + gw / 2.0f, gy = y1 + gh / 2.0f;
    int l = 0;
    while (l < 2 && n >= a.base[l + 1]) ++l;
    const int match = 3 * l + (n - a.base[l]) % 3;
    const double fx = (double)gx / (double)a.W * (double)a.fw[l];
    const double fy = (double)gy / (double)a.H * (double)a.fh[l];
    const long long row = (long long)b * a.N + n;
    a.ctr[row * 2 + 0] = (float)(fx - (double)(int)fx);
    a.ctr[row * 2 + 1] = (float)(fy - (double)(int)fy);
    // max(gtw, 1) is the np.float32 gtw unless 1 > gtw; then it is the Python int and 1 / anchor is float64
    a.scl[row * 2 + 0] = gw >= 1.0f ? vy_logf(gw / kAnchorW[match]) : (float)log(1.0 / (double)kAnchorW[match]);
    a.scl[row * 2 + 1] = gh >= 1.0f ? vy_logf(gh / kAnchorH[match]) : (float)log(1.0 / (double)kAnchorH[match]);
    const float wt = (float)(2.0 - (double)(gw * gh) / (double)a.W / (double)a.H);
    a.wts[row * 2 + 0] = wt;
    a.wts[row * 2 + 1] = wt;
    a.obj[row] = a.gt_mix ? a.gt_mix[(long long)b * a.M + m] : 1.0f;
    const int id = (int)a.gt_ids[(long long)b * a.M + m];
    float* cr = a.cls + row * a.C;
    for (int c = 0; c < a.C; ++c) cr[c] = (c == id) ? 1.0f : 0.0f;
  }
}

}  // namespace

extern "C" int vy_prefetch_targets(const float* gt_boxes, const float* gt_ids, const float* gt_mixratio, int32_t batch,
                                   int32_t num_gt, int32_t height, int32_t width, int32_t num_class, float* objness_t,
                                   float* centers_t, float* scales_t, float* weights_t, float* clas_t, void* stream) {
  if (!objness_t || !centers_t || !scales_t || !weights_t || !clas_t || batch < 1 || num_gt < 0 || num_class < 1 ||
      (num_gt > 0 && (!gt_boxes || !gt_ids)))
    return fail(VY_ERR_INVALID, "bad argument");
  if (height < 32 || width < 32 || height % 32 || width % 32)
    return fail(VY_ERR_INVALID, "height and width must be positive multiples of 32");
  if (num_gt > kMaxGt) return fail(VY_ERR_UNSUPPORTED, "more than %d gt rows per image", kMaxGt);
  TargetArgs a;
  a.gt_boxes = gt_boxes;
  a.gt_ids = gt_ids;
  a.gt_mix = gt_mixratio;
  a.obj = objness_t;
  a.ctr = centers_t;
  a.scl = scales_t;
  a.wts = weights_t;
  a.cls = clas_t;
  a.B = batch;
  a.M = num_gt;
  a.H = height;
  a.W = width;
  a.C = num_class;
  int n = 0;
  const int strides[3] = {32, 16, 8};
  for (int i = 0; i < 3; ++i) {
    a.fh[i] = height / strides[i];
    a.fw[i] = width / strides[i];
    a.base[i] = n;
    n += 3 * a.fh[i] * a.fw[i];
  }
  a.N = n;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long long total = (long long)batch * n * (7 + num_class);
  long long blocks = (total + 256 * 8 - 1) / (256 * 8);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(targets_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
  HIP_TRY(hipGetLastError());
  if (num_gt > 0) {
    hipLaunchKernelGGL(targets_scatter_kernel, dim3(batch), dim3(64), 0, s, a);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}
