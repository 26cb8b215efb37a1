// kernels.h — internal launch interface between the host planner (net.cpp) and the HIP kernels.
// Activation storage ("plane"): zero-bordered NHWC fp32, dims (B, H+2, W+2, C); element (b,y,x,c)
// lives at ((b*(H+2) + y+1)*(W+2) + x+1)*C + c.  Borders are zeroed once when the workspace is
// bound and never written, so 3x3 taps need no bounds checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One launch of the implicit-GEMM conv kernel.  Rows m = (b, y, x) over a logical LH x LW grid;
// the A row of pixel m for tap t is Kc contiguous floats at the plane position
// (y*a_s + a_oy + tap_dy[t], x*a_s + a_ox + tap_dx[t]) (padded coordinates).  The same kernel runs
//   forward      A = input activations, W rows = Cout ([n][k] operand), taps = (kh-p, kw-p)
//   dgrad        A = dz (gradient of the conv output), W read as [k = cout][n = cin], taps flipped;
//                stride-2 convs run one launch per input-pixel parity class (o_s = 2)
struct ConvArgs {
  const float* in;      // A operand plane base
  const float* w;       // weights [Cout][taps][Cin]
  const float* scale;   // per-N, nullable (folded BN scale)
  const float* shift;   // per-N, nullable (folded BN shift, or conv bias when scale == null)
  const float* res;     // addend plane (residual / gradient accumulate), nullable; pixel map of `out`
  float* out;           // output plane base
  float* stats;         // nullable: per-M-tile column sums [tiles_m][2][N] (train-mode BatchNorm)
  int B, LH, LW, M;     // M = B*LH*LW
  int a_Hp, a_Wp, a_cs, a_co, a_s, a_oy, a_ox;
  int Kc;               // A channels per tap, multiple of 32
  int ntaps;
  signed char tap_dy[9], tap_dx[9];
  unsigned char tap_w[9];   // weight tap index (kh*k + kw) of tap t
  int w_taps, w_cin, w_cout;  // weight tensor dims
  int N;                // GEMM columns of this launch
  int o_Hp, o_Wp, o_cs, o_co, o_s, o_oy, o_ox;
  int ups;              // 1, or 2: nearest x2 replicate on store (layers.py:11-20 fused)
  int r_cs, r_co;
  int leaky;            // LeakyReLU(0.1) after the affine
  int dgrad;            // 0: W is the [n][k] operand (forward); 1: W is the [k][n] operand (dgrad)
};

// 3x3/1x1 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32.
hipError_t vy_launch_conv_igemm(const ConvArgs& a, hipStream_t s);
int vy_conv_tiles_m(const ConvArgs& a);

// stem: 3x3 stride-1 conv from the caller's NCHW image (Cin = 3) into a plane, fused affine+leaky.
struct StemArgs {
  const float* x;       // (B,3,H,W) NCHW
  const float* w;       // [Cout][3][3][3]  (kh,kw,cin)
  const float* scale;
  const float* shift;
  float* out;           // plane (B,H+2,W+2,out_cs)
  int B, H, W, Cout, out_cs, out_co;
};
hipError_t vy_launch_stem(const StemArgs& a, hipStream_t s);

// BN fold: scale = gamma / sqrt(var + eps), shift = beta - mean*scale for n_layers BN layers.
struct FoldDesc {
  int64_t gamma, beta, mean, var;   // element offsets into the parameter buffer
  int64_t scale, shift;             // element offsets into the parameter buffer (scratch region)
  int32_t C, pad;
};
hipError_t vy_launch_bn_fold(float* params, const FoldDesc* descs_dev, int n_layers, int max_c,
                             float eps, hipStream_t s);

// plane view -> dense NCHW copy (parity taps)
hipError_t vy_launch_plane_to_nchw(const float* plane, int B, int H, int W, int cs, int co, int C,
                                   float* dst, hipStream_t s);

// ---- detection tail -------------------------------------------------------------------------
#define VY_NMS_MAX_TOPK 1024
struct HeadView {
  const float* pred;    // plane holding the prediction conv output, channels a*(5+C)+p
  int H, W, cs, co;     // unpadded spatial, plane channel stride / offset
  float stride;         // 32, 16, 8
  float aw[3], ah[3];   // anchors (pixels)
  int cand_base;        // first candidate index of this scale in the reference's concat order
};
struct DetArgs {
  HeadView head[3];
  int B, C;             // batch, classes
  int n_cand;           // N*C candidates per image
  float valid_thresh, nms_thresh;
  int topk, post_nms;
  int do_nms;           // 0: nms disabled (nms_thresh outside (0,1)): return first post_nms rows
};
size_t vy_det_scratch_bytes(int B);
// full tail: decode -> radix select of the top-k valid scores -> sort -> per-class NMS -> outputs
hipError_t vy_launch_detect(const DetArgs& a, void* scratch, float* ids, float* scores, float* bboxes,
                            int32_t* keep_idx, hipStream_t s);
