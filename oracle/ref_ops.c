/*
 * oracle/ref_ops.c — CPU restatement of the tensor operators on the yolo3_darknet53 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in videoyolo_amd/ may include, link or call this file;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * PARITY UNPINNED: the reference delegates every operator below to mxnet / gluoncv, which are
 * neither vendored in /root/reference nor installable here (requirements.txt:1-2 lists them
 * unversioned).  The reference holds no tests or golden vectors for this path.
 * These functions restate the operators' *published* semantics at the reference's call sites;
 * they have not been checked against a running mxnet.  What pins them instead: hand-derived
 * known-answer cases (tests/test_oracle_known_answers.py) and an independent torch-CPU
 * cross-check (tests/test_oracle_vs_torch.py).
 *
 * Layouts are the reference's: activations NCHW, conv weights OIHW, fp32 everywhere.
 *
 * Summation order (documented because the HIP kernels are held to bit-equality with it):
 *   the products of an output element are visited over taps (kh, kw) with kh outermost and, inside a tap, over
 *   the input channels; when Cin is a multiple of 8 the channels of each aligned group of 8 are visited in
 *   the order 0,4,1,5,2,6,3,7 (the order in which the fp32 matrix instruction of the device consumes a
 *   32-byte channel group: its two half-waves hold channels 0-3 and 4-7 and alternate), otherwise (the
 *   3-channel stem) ascending.  That sequence, counted in k-steps of one tap x 32 channels, is cut into
 *   S = vy_conv_k_chunks(taps * 32 * ceil(Cin / 32)) contiguous runs of equal length (include/vy_math.h:
 *   4 for K >= 4096 — the 3x3 cells on 512 channels —, else 1); every run is ONE fp32 fma chain starting from +0, and the
 *   runs are added in order starting from +0:  out = (((+0 + P0) + P1) + P2) + P3.  (Rounds 1-5 used a single
 *   chain for every layer; round 6 cut the long ones so that one frame's small maps can use the whole chip.)
 *   A dot product's summation order is not part of the reference's definition (mxnet delegates it to
 *   MKL-DNN / cuDNN); fixing one order on both sides is what makes bit-equality testable.
 *   Bias/BN/activation follow.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/vy_math.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* bench.py's cpu_baseline: as many threads as the host really grants (a cgroup CPU quota below the core count makes the
 * default — one thread per logical CPU — thrash) */
void vyo_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int vyo_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------------------------------------------------------------------------------------
 * Conv2D, cross-correlation, zero padding.   reference: nn.Conv2D at layers.py:66-67 (no bias)
 * and yolo3.py:62 (1x1, bias).   x (N,C,H,W), w (O,C,k,k), y (N,O,Ho,Wo), Ho = (H+2p-k)/s + 1.
 * Optional per-output-channel epilogue, applied in this order (any pointer may be NULL):
 *   y = fma(acc, scale[o], shift[o])     (folded eval-mode BatchNorm, or scale=NULL, shift=bias:
 *                                         y = acc + bias[o])
 *   y = leaky(y)                         if leaky != 0         (layers.py:69, slope 0.1)
 * ------------------------------------------------------------------------------------------- */
void vyo_conv2d(const float* x, int N, int C, int H, int W, const float* w, int O, int k, int s,
                int p, const float* scale, const float* shift, int leaky, float* y) {
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  /* the runs of the pinned order (header): k-step t = tap * cch + (channel / 32), S runs of T / S k-steps */
  const int cch = (C + 31) / 32, T = k * k * cch;
  const int S = vy_conv_runs(k * k, cch);
  const int run = T / S; /* T = 144 wherever S > 1 */
#pragma omp parallel
  {
    float* acc = (float*)malloc(sizeof(float) * (size_t)Wo);
    float* tot = (float*)malloc(sizeof(float) * (size_t)Wo);
#pragma omp for collapse(3) schedule(static)
    for (int n = 0; n < N; ++n)
      for (int o = 0; o < O; ++o)
        for (int oy = 0; oy < Ho; ++oy) {
          for (int ox = 0; ox < Wo; ++ox) acc[ox] = 0.0f, tot[ox] = 0.0f;
          int cur = 0; /* the run `acc` is the chain of */
          for (int kh = 0; kh < k; ++kh) {
            const int iy = oy * s + kh - p;
            if (iy < 0 || iy >= H) continue; /* zero taps: fma(0, w, acc) == acc */
            for (int kw = 0; kw < k; ++kw) {
              /* valid ox range for this tap: 0 <= ox*s + kw - p < W */
              int lo = 0, hi = Wo;
              while (lo < Wo && lo * s + kw - p < 0) ++lo;
              while (hi > lo && (hi - 1) * s + kw - p >= W) --hi;
              for (int cc = 0; cc < C; ++cc) {
                if (S > 1 && (cc & 31) == 0) {
                  /* a new k-step: close every run that ends before it (runs skipped entirely — all their taps fall
                   * into the zero padding — contribute a chain that is still +0) */
                  const int r = ((kh * k + kw) * cch + (cc >> 5)) / run;
                  for (; cur < r; ++cur)
                    for (int ox = 0; ox < Wo; ++ox) tot[ox] = tot[ox] + acc[ox], acc[ox] = 0.0f;
                }
                /* channel visiting order, see the header: 0,4,1,5,2,6,3,7 per group of 8 */
                const int c = (C % 8 == 0) ? ((cc & ~7) | (((cc & 1) << 2) | ((cc & 7) >> 1))) : cc;
                const float wv = w[(((size_t)o * C + c) * k + kh) * k + kw];
                const float* xr = x + (((size_t)n * C + c) * H + iy) * W + (kw - p);
#pragma omp simd
                for (int ox = lo; ox < hi; ++ox) acc[ox] = fmaf(xr[ox * s], wv, acc[ox]);
              }
            }
          }
          if (S > 1) /* the remaining runs (the last one always): (((+0 + P0) + P1) + P2) + P3 */
            for (; cur < S; ++cur)
              for (int ox = 0; ox < Wo; ++ox) tot[ox] = tot[ox] + acc[ox], acc[ox] = 0.0f;
          float* yr = y + (((size_t)n * O + o) * Ho + oy) * Wo;
          const float sc = scale ? scale[o] : 1.0f;
          const float sh = shift ? shift[o] : 0.0f;
          for (int ox = 0; ox < Wo; ++ox) {
            float v = S > 1 ? tot[ox] : acc[ox];
            if (scale)
              v = fmaf(v, sc, sh);
            else if (shift)
              v = v + sh;
            if (leaky) v = vy_leaky(v);
            yr[ox] = v;
          }
        }
    free(acc);
    free(tot);
  }
}

/* Eval-mode BatchNorm folded to (scale, shift); reference: norm_layer(epsilon=1e-5, ...) at
 * layers.py:68 run outside autograd.train_mode -> uses running_mean / running_var. */
void vyo_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                 float eps, int C, float* scale, float* shift) {
  for (int c = 0; c < C; ++c) {
    scale[c] = vy_bn_scale(gamma[c], var[c], eps);
    shift[c] = vy_bn_shift(beta[c], mean[c], scale[c]);
  }
}

/* Train-mode BatchNorm forward over (N,H,W) per channel + LeakyReLU.  [UPSTREAM-RECALLED]
 * mxnet BatchNorm(fix_gamma=False, use_global_stats=False): mean = sum/count, var = biased
 * variance, y = (x-mean)/sqrt(var+eps)*gamma+beta.  Sums are accumulated in double and rounded to
 * fp32 mean / var (the device does the same, so both sides normalise with identical constants).
 * Outputs the batch mean/var (biased) so the caller can update running stats. */
void vyo_bn_train(const float* x, int N, int C, int HW, const float* gamma, const float* beta,
                  float eps, int leaky, float* y, float* mean_out, float* var_out) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < C; ++c) {
    double s = 0, ss = 0;
    for (int n = 0; n < N; ++n) {
      const float* xr = x + ((size_t)n * C + c) * HW;
      for (int i = 0; i < HW; ++i) {
        s += xr[i];
        ss += (double)xr[i] * xr[i];
      }
    }
    const double cnt = (double)N * HW;
    const double m = s / cnt;
    double v = ss / cnt - m * m;
    if (v < 0) v = 0;
    const float mf = (float)m, vf = (float)v;
    mean_out[c] = mf;
    var_out[c] = vf;
    /* applied as ONE fma with scale = gamma*invstd, shift = beta - mean*scale (the device's form;
     * differs from the textbook (x-mean)*invstd*gamma+beta by rounding only) */
    const float inv = 1.0f / sqrtf(vf + eps);
    const float sc = gamma[c] * inv;
    const float sh = fmaf(-mf, sc, beta[c]);
    for (int n = 0; n < N; ++n) {
      const float* xr = x + ((size_t)n * C + c) * HW;
      float* yr = y + ((size_t)n * C + c) * HW;
      for (int i = 0; i < HW; ++i) {
        float t = fmaf(xr[i], sc, sh);
        yr[i] = leaky ? vy_leaky(t) : t;
      }
    }
  }
}

/* elementwise helpers used by the numpy graph code so that exp/sigmoid/log are the
 * reproducible vy_math versions (yolo3.py:172-175) */
void vyo_sigmoid(const float* x, size_t n, float* y) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) y[i] = vy_sigmoidf(x[i]);
}
void vyo_exp(const float* x, size_t n, float* y) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) y[i] = vy_expf(x[i]);
}
void vyo_log(const float* x, size_t n, float* y) {
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; ++i) y[i] = vy_logf(x[i]);
}

/* ---------------------------------------------------------------------------------------------
 * contrib.box_nms as called at yolo3.py:1198-1200:
 *   box_nms(data (B,n,6), overlap_thresh, valid_thresh, topk, id_index=0, score_index=1,
 *           coord_start=2, force_suppress=False)            in_format = out_format = 'corner'
 * [UPSTREAM-RECALLED, UNVERIFIED] semantics restated from mxnet's operator documentation:
 *   1. rows with score <= valid_thresh are invalid (background_id = -1: none excluded by id);
 *   2. valid rows are sorted by score, descending; ties keep ascending input order (stable);
 *   3. only the first topk sorted rows take part (topk <= 0: all of them);
 *   4. for i < j in that order, j is suppressed when i survives, id_i == id_j (force_suppress
 *      False) and IoU(i, j) > overlap_thresh;
 *   5. the output has the input's shape: surviving rows first, in sorted order, every other
 *      row filled with -1.
 * out_index (nullable, (B,n) int32): for each output row the input row index it came from, -1
 * for filler — this is the "box indices" stream the parity tests compare exactly.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  float score;
  int32_t idx;
} vyo_key;

static int vyo_key_cmp(const void* a, const void* b) {
  const vyo_key* x = (const vyo_key*)a;
  const vyo_key* y = (const vyo_key*)b;
  if (x->score > y->score) return -1;
  if (x->score < y->score) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx);
}

void vyo_box_nms(const float* data, int B, int n, float overlap_thresh, float valid_thresh,
                 int topk, int force_suppress, float* out, int32_t* out_index) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < B; ++b) {
    const float* d = data + (size_t)b * n * 6;
    float* o = out + (size_t)b * n * 6;
    int32_t* oi = out_index ? out_index + (size_t)b * n : NULL;
    vyo_key* keys = (vyo_key*)malloc(sizeof(vyo_key) * (size_t)(n > 0 ? n : 1));
    int nv = 0;
    for (int i = 0; i < n; ++i)
      if (d[(size_t)i * 6 + 1] > valid_thresh) {
        keys[nv].score = d[(size_t)i * 6 + 1];
        keys[nv].idx = i;
        ++nv;
      }
    qsort(keys, (size_t)nv, sizeof(vyo_key), vyo_key_cmp);
    int m = (topk > 0 && topk < nv) ? topk : nv;
    char* dead = (char*)calloc((size_t)(m > 0 ? m : 1), 1);
    for (int i = 0; i < m; ++i) {
      if (dead[i]) continue;
      const float* bi = d + (size_t)keys[i].idx * 6;
      for (int j = i + 1; j < m; ++j) {
        if (dead[j]) continue;
        const float* bj = d + (size_t)keys[j].idx * 6;
        if (!force_suppress && bi[0] != bj[0]) continue;
        float iou = vy_box_iou(bi[2], bi[3], bi[4], bi[5], bj[2], bj[3], bj[4], bj[5]);
        if (iou > overlap_thresh) dead[j] = 1;
      }
    }
    int cnt = 0;
    for (int i = 0; i < m; ++i) {
      if (dead[i]) continue;
      memcpy(o + (size_t)cnt * 6, d + (size_t)keys[i].idx * 6, 6 * sizeof(float));
      if (oi) oi[cnt] = keys[i].idx;
      ++cnt;
    }
    for (int i = cnt; i < n; ++i) {
      for (int q = 0; q < 6; ++q) o[(size_t)i * 6 + q] = -1.0f;
      if (oi) oi[i] = -1;
    }
    free(dead);
    free(keys);
  }
}

/* gluoncv BBoxBatchIOU (yolo_target.py:171,202): a (B,N,4), b (B,M,4) corner format, offset 0,
 * eps 1e-15 -> iou (B,N,M).  [UPSTREAM-RECALLED] i = max(0,w)*max(0,h); iou = i/(aa+ab-i+eps). */
void vyo_batch_iou(const float* a, const float* b, int B, int N, int M, float* out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < B; ++bi)
    for (int i = 0; i < N; ++i) {
      const float* p = a + ((size_t)bi * N + i) * 4;
      const float aa = (p[2] - p[0]) * (p[3] - p[1]);
      for (int j = 0; j < M; ++j) {
        const float* q = b + ((size_t)bi * M + j) * 4;
        float iw = fminf(p[2], q[2]) - fmaxf(p[0], q[0]);
        float ih = fminf(p[3], q[3]) - fmaxf(p[1], q[1]);
        iw = iw > 0.0f ? iw : 0.0f;
        ih = ih > 0.0f ? ih : 0.0f;
        const float inter = iw * ih;
        const float ab = (q[2] - q[0]) * (q[3] - q[1]);
        out[((size_t)bi * N + i) * M + j] = inter / (((aa + ab) - inter) + 1e-15f);
      }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Backward of Conv2D (what mxnet autograd computes for layers.py:66 / yolo3.py:62), written from
 * the definition  y[n,o,oy,ox] = sum_{c,kh,kw} x[n,c,oy*s+kh-p,ox*s+kw-p] * w[o,c,kh,kw]:
 *   dx[n,c,iy,ix] = sum_{o,kh,kw : iy = oy*s+kh-p, ix = ox*s+kw-p} dy[n,o,oy,ox] * w[o,c,kh,kw]
 *   dw[o,c,kh,kw] = sum_{n,oy,ox} dy[n,o,oy,ox] * x[n,c,oy*s+kh-p,ox*s+kw-p]
 * Accumulated in double: this side is the accuracy reference for the fp32 device kernels.
 * ------------------------------------------------------------------------------------------- */
void vyo_conv2d_bwd_data(const float* dy, int N, int O, int Ho, int Wo, const float* w, int C, int k,
                         int s, int p, int H, int W, float* dx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      double* acc = (double*)calloc((size_t)H * W, sizeof(double));
      for (int o = 0; o < O; ++o)
        for (int kh = 0; kh < k; ++kh)
          for (int kw = 0; kw < k; ++kw) {
            const double wv = w[(((size_t)o * C + c) * k + kh) * k + kw];
            for (int oy = 0; oy < Ho; ++oy) {
              const int iy = oy * s + kh - p;
              if (iy < 0 || iy >= H) continue;
              const float* dr = dy + (((size_t)n * O + o) * Ho + oy) * Wo;
              for (int ox = 0; ox < Wo; ++ox) {
                const int ix = ox * s + kw - p;
                if (ix < 0 || ix >= W) continue;
                acc[(size_t)iy * W + ix] += wv * dr[ox];
              }
            }
          }
      float* out = dx + ((size_t)n * C + c) * H * W;
      for (size_t i = 0; i < (size_t)H * W; ++i) out[i] = (float)acc[i];
      free(acc);
    }
}

void vyo_conv2d_bwd_weight(const float* dy, int N, int O, int Ho, int Wo, const float* x, int C, int k,
                           int s, int p, int H, int W, float* dw) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int o = 0; o < O; ++o)
    for (int c = 0; c < C; ++c)
      for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw) {
          double acc = 0;
          for (int n = 0; n < N; ++n)
            for (int oy = 0; oy < Ho; ++oy) {
              const int iy = oy * s + kh - p;
              if (iy < 0 || iy >= H) continue;
              const float* dr = dy + (((size_t)n * O + o) * Ho + oy) * Wo;
              const float* xr = x + (((size_t)n * C + c) * H + iy) * W;
              for (int ox = 0; ox < Wo; ++ox) {
                const int ix = ox * s + kw - p;
                if (ix < 0 || ix >= W) continue;
                acc += (double)dr[ox] * xr[ix];
              }
            }
          dw[(((size_t)o * C + c) * k + kh) * k + kw] = (float)acc;
        }
}
