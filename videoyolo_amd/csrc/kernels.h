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
// exact n / d for any 32-bit n by one multiply-high (Granlund-Montgomery round-up): t = mulhi(m, n);
// q = (t + ((n - t) >> s1)) >> s2
struct VyFastDiv {
  unsigned m, s1, s2;
};

struct ConvArgs {
  const float* in;      // A operand plane base
  const float* w;       // weights [Cout][taps][Cin]
  const float* scale;   // per-N, nullable (folded BN scale)
  const float* shift;   // per-N, nullable (folded BN shift, or conv bias when scale == null)
  const float* res;     // addend plane (residual / gradient accumulate), nullable; pixel map of `out`
  float* out;           // output plane base
  double* stats;        // nullable: per-M-tile column sums [tiles_m][2][N] (train-mode BatchNorm)
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
  VyFastDiv fd_lw, fd_lh;       // filled by the launcher: division by LW / LH (the row tables of every tile)
  unsigned pk_dy, pk_dx;        // filled by the launcher: tap tables packed 2 bits / tap (value + 1)
#ifdef VY_CONV_TRACE
  unsigned long long* trace;    // tools/probe/conv_tile_trace.hip only: [block][8] phase timestamps (100 MHz) + CU id
#endif
  unsigned long long pk_w;      // 4 bits / tap
  // stream-K scratch of the net's workspace (nullable: plain launches only): partial accumulator slabs + hand-off flags
  float* sk_partials;
  unsigned* sk_flags;
  unsigned long long sk_bytes;  // bytes of sk_partials
  int sk_nflags;
  // conv_split.hip only: this conv's weights as pre-split bf16 tile images (vy_launch_split_weights); nullable
  const void* w_split;
  // ... and scratch for its split-K launches ([ksplit][M][N] fp32 partial sums): the same workspace region as
  // sk_partials (a stream never runs a stream-K exact launch and a split-K launch at once); nullable: no k-split
  float* splitk_slabs;
  unsigned long long splitk_bytes;
  // conv_wino.hip only: this conv's weights as four Winograd-transformed, pre-split image sets (vy_launch_wino_weights);
  // nullable
  const void* w_wino;
  // the test switches VY_SPLIT_ALWAYS / VY_SPLIT_WINO as the net read them ONCE at the start of this forward / step
  // (value + 1; 0: not read — vy_conv_*_pays read the environment themselves, e.g. for a probe's hand-made ConvArgs)
  int env_split_always_p1, env_wino_mode_p1;
  // The pinned summation order cuts long K into S = vy_conv_k_chunks(K) runs (include/vy_math.h).  A workgroup that computes
  // more than one run of a tile parks every finished chain in this scratch ([run][tile][BM x BN] floats: written and read
  // back by the same lanes, or handed from the block that starts a stream-K tile to the one that finishes it,
  // write-through) and adds them in run order at the end.  Nullable when no conv of the net has S > 1; a launch with S > 1
  // and no scratch is refused.  k_chunk: filled by the launcher (k-steps per run; 0: one run)
  float* ck_scratch;
  unsigned long long ck_bytes;
  int k_chunk;
  // CUs of the device this launch goes to, as the net resolved them ONCE (vy_net::cus: at its first sizing call, checked
  // against the workspace's device at bind time) — every cost-model choice of a net counts rounds in the same chip whatever
  // the calling thread's current device is.  0: not set (a probe's hand-made ConvArgs): vy_cu_count() of the current device
  int cus;
};
#define VY_SK_PARTIAL_BYTES (32u << 20)  // 512 blocks x 128x128 fp32 (the largest instance: 2 blocks per CU x 256 CUs)
#define VY_SK_FLAGS 2048

// 3x3/1x1 implicit-GEMM convolution on v_mfma_f32_32x32x2_f32.
hipError_t vy_launch_conv_igemm(const ConvArgs& a, hipStream_t s);
// the same on 16x16 wave tiles (v_mfma_f32_16x16x4_f32; conv_small.hip): forward launches too small to fill the chip
// with 32x32 wave tiles.  Block tile bm x bn = 32x32 or 32x64; called by vy_launch_conv_igemm.
hipError_t vy_launch_conv_s16(const ConvArgs& a, int bm, int bn, hipStream_t s);
int vy_conv_tiles_m(const ConvArgs& a);
void vy_conv_cfg(const ConvArgs& a, int* bm, int* bn);  // block tile the launch will use
double vy_conv_predict_us(const ConvArgs& a);             // the cost model's time for the launch (conv_cost_model.h)
int vy_cu_count();                                        // CUs of the CURRENT device (nets resolve theirs once: vy_net::cus)
int vy_cu_count_of(int device);                           // ... of a given device ordinal
int vy_cu_count_of_ptr(const void* dev_ptr);              // ... of the device that owns an allocation (0: not a device pointer)
static inline int vy_args_cus(const ConvArgs& a) { return a.cus > 0 ? a.cus : vy_cu_count(); }
bool vy_conv_streamk(const ConvArgs& a);                  // ... and whether it will be a stream-K launch (label "<BM>x<BN>sk")
int vy_conv_ksplit(const ConvArgs& a);                    // ... or a split-K one: the S workgroups per tile (label "<BM>x<BN>ks<S>"), else 0
size_t vy_conv_chunk_scratch_bytes(long long M, int N, int runs);   // ck_scratch bytes a conv of M x N outputs summed in `runs` runs may need (any tile)

// stream-K is enabled per net only after this has seen the MI355X's SPX placement (8 XCDs, blocks L and L + 8 on one
// XCD, 256 CUs) on the current device; `scratch_dev`: >= 512 words the probe may use and leaves zeroed.  Synchronises `s`
// the first time it is called for a device; 0 while `s` is being captured.
int vy_sk_verify_topology(unsigned* scratch_dev, hipStream_t s);

// Opt-in split-fp32 forward conv on the bf16 matrix core (conv_split.hip: bf16 x 3, six products, fp32 accumulate).
// Same ConvArgs / planes as vy_launch_conv_igemm, plus a.w_split = the conv's weights as bf16 tile images.
size_t vy_split_weight_bytes(int cout, int taps, int cin);
hipError_t vy_launch_split_weights(const float* w, void* img, int cout, int taps, int cin, hipStream_t s);
// ... and as the [k = cout][n = cin] operand of the data gradient (cin % 32 == 0; cout zero-padded to 32)
size_t vy_split_weight_dgrad_bytes(int cout, int taps, int cin);
hipError_t vy_launch_split_weights_dgrad(const float* w, void* img, int cout, int taps, int cin, hipStream_t s);
// all images of a net in one launch: descriptors (device memory, sorted by `first`) of `n` image sets covering `total`
// 8-channel OCTETS (a thread converts one; dgrad sets count their zero-padded cout); params / ws: the parameter buffer
// and the workspace
struct SplitDesc {
  long long first;     // index of this set's first octet in the launch's flat octet range
  long long w_off;     // element offset of the conv's weights in the parameter buffer
  long long img_off;   // byte offset of the images in the workspace
  int cout, taps, cin, dgrad;
};
hipError_t vy_launch_split_weights_batch(const float* params, void* ws, const SplitDesc* descs_dev, int n, long long total,
                                         hipStream_t s);
// ... and, for its long-K 3x3 stride-1 cells, as a 1-D Winograd F(2, 3) on the same split arithmetic (conv_wino.hip;
// a.w_wino = the transformed image sets)
size_t vy_wino_weight_bytes(int cout, int cin);
hipError_t vy_launch_wino_weights(const float* w, void* img, int cout, int cin, hipStream_t s);
bool vy_conv_wino_supported(const ConvArgs& a);
bool vy_conv_wino_pays(const ConvArgs& a);          // the cost models' verdict (conv_cost_model.h: vy_predict_wino); VY_SPLIT_WINO=0 / 2
hipError_t vy_launch_conv_wino(const ConvArgs& a, hipStream_t s);
bool vy_conv_split_supported(const ConvArgs& a);   // forward, N % 64 == 0, Kc % 32 == 0, an epilogue the kernel has
void vy_conv_split_cfg(const ConvArgs& a, int* bm, int* bn, int* ksplit);  // block tile and k-split the launch will use
// conv mode VY_CONV_SPLIT_BF16X3, per launch: supported AND predicted faster than the exact kernel (small launches —
// a single frame's deep layers — stay on the exact kernel, which has the small tiles and stream-K)
bool vy_conv_split_pays(const ConvArgs& a);
hipError_t vy_launch_conv_split(const ConvArgs& a, hipStream_t s);

// stem: 3x3 stride-1 conv from the caller's NCHW image (Cin = 3) into a plane, fused affine+leaky.
// BN fold: scale = gamma / sqrt(var + eps), shift = beta - mean*scale for n_layers BN layers.
struct FoldDesc {
  int64_t gamma, beta, mean, var;   // element offsets into the parameter buffer
  int64_t scale, shift;             // element offsets into the parameter buffer (scratch region)
  int32_t C, pad;
};
struct StemArgs {
  const float* x;       // (B,3,H,W) NCHW
  const float* w;       // [Cout][3][3][3]  (kh,kw,cin)
  const float* scale;
  const float* shift;
  float* out;           // plane (B,H+2,W+2,out_cs)
  int B, H, W, Cout, out_cs, out_co;
  // Inference only, optional: the eval-mode BatchNorm fold of the WHOLE net rides in this launch (the first of a forward)
  // instead of a launch of its own in front of it — block i < fold_n folds layer i exactly as bn_fold_kernel does, and every
  // lane takes the stem's own affine straight from its BatchNorm parameters (layer fold_stem; scale / shift are then not
  // read).  8.5 us off the chain of one frame.  vy_launch_stem refuses (hipErrorInvalidValue) when the launch has fewer
  // blocks than layers: ask vy_stem_can_fold first.
  float* fold_params = nullptr;
  const FoldDesc* fold_descs = nullptr;
  int fold_n = 0, fold_stem = 0;
  float fold_eps = 0.0f;
};
hipError_t vy_launch_stem(const StemArgs& a, hipStream_t s);
bool vy_stem_can_fold(int B, int H, int W, int n_layers);   // the inference stem launch of this shape has >= n_layers blocks

// (FoldDesc: above StemArgs)
hipError_t vy_launch_bn_fold(float* params, const FoldDesc* descs_dev, int n_layers, int max_c,
                             float eps, hipStream_t s);

// plane view -> dense NCHW copy (parity taps)
hipError_t vy_launch_plane_to_nchw(const float* plane, int B, int H, int W, int cs, int co, int C,
                                   float* dst, hipStream_t s);
hipError_t vy_launch_nchw_to_plane(const float* src, int B, int H, int W, int cs, int co, int C, float* plane,
                                   hipStream_t s);

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
size_t vy_det_scratch_bytes(int B, int n_items, int C);  // n_items = anchors per image (N), C = classes
// full tail: decode -> radix select of the top-k valid scores -> sort -> per-class NMS -> outputs
// nms off: the full (B, N*C, .) detection tensor in the reference's row order
hipError_t vy_launch_raw_detections(const DetArgs& a, float* ids, float* scores, float* bboxes, int32_t* keep_idx,
                                    hipStream_t s);
hipError_t vy_launch_detect(const DetArgs& a, void* scratch, float* ids, float* scores, float* bboxes,
                            int32_t* keep_idx, hipStream_t s);

// =============================================================================================
// Training kernels (train_kernels.hip, wgrad.hip)
// =============================================================================================
// out[c] = sum over t < n_part of partials[t*n_cols + c], accumulated in double in index order
// (deterministic).  Used for BatchNorm statistics, BN-backward sums, bias gradients.
hipError_t vy_launch_reduce_partials(const float* partials, int n_part, int n_cols, double* out,
                                     hipStream_t s);
hipError_t vy_launch_reduce_partials_f64(const double* partials, int n_part, int n_cols, double* out,
                                         hipStream_t s);

hipError_t vy_launch_f64_to_f32(const double* src, float* dst, int n, hipStream_t s);

// batch statistics -> normalisation coefficients (mxnet BatchNorm, train mode; layers.py:68)
struct BnFinalizeArgs {
  const double* sums;     // [2][C]: sum x, sum x^2 over `count` samples (all ranks when SyncBN)
  double count;
  const float* gamma;
  const float* beta;
  float* running_mean;    // updated in place: r = momentum*r + (1-momentum)*batch
  float* running_var;
  float* scale;           // gamma * invstd
  float* shift;           // beta - mean*scale
  float* save_mean;       // for backward
  float* save_invstd;
  int C;
  float eps, momentum;
};
hipError_t vy_launch_bn_finalize(const BnFinalizeArgs& a, hipStream_t s);
// ordered reduce of the [n_part][2][C] partial sums + finalize (per-device BatchNorm): one launch, or — more than
// 2 * VY_REDUCE_SLICES rows and a scratch of VY_REDUCE_SLICES * 2 * C doubles given — a slice-sum launch first
#define VY_REDUCE_SLICES 64
hipError_t vy_launch_bn_reduce_finalize(const double* partials, int n_part, const BnFinalizeArgs& a, double* scratch,
                                        hipStream_t s);

// a = leaky(fma(z, scale, shift)) (+ res), z plane (B,H+2,W+2,C) -> output view (x1 or x2 replicate)
struct BnApplyArgs {
  const float* z;
  const float* scale;
  const float* shift;
  const float* res;       // nullable, pixel map of out
  float* out;
  int B, H, W, C;         // z plane: cs == C, co == 0
  int o_Hp, o_Wp, o_cs, o_co, ups;
  int r_cs, r_co;
};
hipError_t vy_launch_bn_apply(const BnApplyArgs& a, hipStream_t s);

// BatchNorm + LeakyReLU backward.  da is read from a gradient view (x2-replicated outputs are
// summed over their 2x2 replicas); y = fma(z, scale, shift) decides the leaky branch.
struct BnBwdArgs {
  const float* g;         // gradient plane of the cell's output view
  float* z;               // raw conv output; overwritten with dz by the apply pass
  const float* scale;
  const float* shift;
  const float* save_mean;
  const float* save_invstd;
  const float* coef;      // [3][C]: c1 = gamma*invstd, c2 = dbeta/n, c3 = dgamma/n  (apply pass)
  float* partials;        // [n_chunks][2][C]  (reduce pass)
  int B, H, W, C;
  int g_Hp, g_Wp, g_cs, g_co, ups;
  int chunk;              // image rows (b, y) per partial chunk: vy_bn_bwd_rows_per_chunk
};
int vy_bn_bwd_rows_per_chunk(int B, int H, int C);
int vy_bn_bwd_chunks(const BnBwdArgs& a);
hipError_t vy_launch_bn_bwd_reduce(const BnBwdArgs& a, hipStream_t s);
void vy_bn_prio_init();   // reads VY_BN_PRIO once (train_kernels.hip: issue priority of the BatchNorm passes)
struct BnBwdFinalizeArgs {
  const double* sums;     // [2][C]: sum dy, sum dy*xhat (all ranks when SyncBN)
  double count;
  const float* gamma;
  const float* save_invstd;
  float* dgamma;          // gradient buffer slots
  float* dbeta;
  float* coef;            // [3][C]
  int C;
  int local_only;         // SyncBN: dgamma/dbeta stay LOCAL sums (they are all-reduced with the grads);
  const double* local_sums;
};
hipError_t vy_launch_bn_bwd_finalize(const BnBwdFinalizeArgs& a, hipStream_t s);
hipError_t vy_launch_bn_bwd_reduce_finalize(const float* partials, int n_part, const BnBwdFinalizeArgs& a, hipStream_t s);
hipError_t vy_launch_bn_bwd_apply(const BnBwdArgs& a, hipStream_t s);

// per-channel sums of a plane view over all pixels -> partials [n_chunks][C] (prediction-conv bias grad)
hipError_t vy_launch_colsum(const float* plane, int B, int H, int W, int cs, int co, int C, int chunk,
                            float* partials, hipStream_t s);
int vy_colsum_chunks(int B, int H, int W, int chunk);

// ---- weight gradient: dW[o][tap][cin] = sum_p dz[p][o] * a[p*s + tap][cin]  (split-K slabs)
struct WgradArgs {
  const float* dz;        // plane (B, Ho+2, Wo+2, z_cs)
  const float* a;         // input activation view
  float* slabs;           // [splits][Cout][taps*Cin]
  const float* zero;      // >= 512 zero bytes (unused since the pixel table; kept for the argument layout)
  const uint2* tab;       // pixel table of this conv (vy_launch_wgrad_table): [M rounded up to 32, + 32] entries
  int B, Ho, Wo, M;       // M = B*Ho*Wo
  int z_cs, Cout;
  int a_Hp, a_Wp, a_cs, a_co, stride;
  int k, Cin;             // kernel size, input channels (multiple of 32)
  int splits, k_per_split;  // k_per_split pixels (multiple of 32) per split
  int xcd_order;          // filled by the launcher: XCD-contiguous (split, tile) order (wgrad.hip)
};
hipError_t vy_launch_wgrad(const WgradArgs& a, hipStream_t s);
// opt-in split-fp32 weight gradient (wgrad_split.hip: bf16 x 3, six products; same WgradArgs, table and slabs):
// Cout % 128 == 0
bool vy_wgrad_split_supported(const WgradArgs& a);
hipError_t vy_launch_wgrad_split(const WgradArgs& a, hipStream_t s);
// output rows (dz channels) of a weight-gradient block tile: 64 or 128 (the planner's split count depends on it)
int vy_wgrad_tile_rows(int Cout, int k, int Cin);
// per-pixel byte offsets (dz vector, input-plane centre pixel shifted by one row + one column) for vy_launch_wgrad,
// relative to the first pixel of the pixel's split (k_per_split: the WgradArgs value the table will be used with);
// n_entries = vy_wgrad_table_entries(M).  Planes may be of any size (round 3: < 4 GiB).
inline size_t vy_wgrad_table_entries(long long M) { return (size_t)((M + 31) / 32 * 32 + 32); }
hipError_t vy_launch_wgrad_table(void* tab, int M, int n_entries, int Ho, int Wo, int z_cs, int a_Hp, int a_Wp, int a_cs,
                                 int stride, int B, int k_per_split, hipStream_t s);
// dst[i] = sum_s slabs[s][i]  (fixed order)
hipError_t vy_launch_slab_reduce(const float* slabs, int splits, long long n, float* dst, hipStream_t s);

// stem weight gradient (Cin = 3): partials [blocks][32*27] then vy_launch_reduce_partials
struct StemWgradArgs {
  const float* x;         // (B,3,H,W) NCHW image
  const float* dz;        // plane (B,H+2,W+2,32)
  float* partials;
  int B, H, W;
};
int vy_stem_wgrad_blocks(int B, int H, int W);
hipError_t vy_launch_stem_wgrad(const StemWgradArgs& a, hipStream_t s);

// stem forward in training: raw conv -> z plane + per-block column sums [blocks][2][32]
int vy_stem_blocks(int B, int H, int W);
hipError_t vy_launch_stem_raw(const StemArgs& a, double* partials, hipStream_t s);

// ---- fused targets + loss + d(loss)/d(raw predictions)
// yolo_target.py:173-205 (dynamic ignore mask), :226-281 (merge), gluoncv YOLOV3Loss (yolo3.py:1187)
struct LossArgs {
  HeadView head[3];       // prediction planes (read)
  float* dpred[3];        // gradient planes, same geometry as head[i].pred (written)
  const float* gt_boxes;  // (B, M, 4) corner, -1 padded
  const float* obj_t;     // (B, N, 1)
  const float* centers_t; // (B, N, 2)
  const float* scales_t;  // (B, N, 2)
  const float* weights_t; // (B, N, 2)
  const float* clas_t;    // (B, N, C)
  float* partials;        // [blocks_per_image][B][4]
  int B, C, M, N;
  float ignore_iou_thresh;
  int label_smooth;
};
int vy_loss_blocks_per_image(int N);
hipError_t vy_launch_loss(const LossArgs& a, hipStream_t s);
// losses[l][b] = sum over blocks of partials (fixed order), l = obj, center, scale, cls
hipError_t vy_launch_loss_reduce(const float* partials, int blocks_per_image, int B, float* losses,
                                 hipStream_t s);

// ---- train-mode non-recording outputs (yolo3.py:1189-1192, items 0 and 4-7 of the 8-tuple)
struct RawPredArgs {
  HeadView head[3];
  float* box;         // (B, N, 4) decoded corner boxes
  float* centers;     // (B, N, 2) raw
  float* scales;      // (B, N, 2) raw
  float* objness;     // (B, N, 1) raw
  float* class_pred;  // (B, N, C) raw
  int B, C, N;
};
hipError_t vy_launch_raw_preds(const RawPredArgs& a, hipStream_t s);

// ---- SGD with momentum (mx.optimizer.SGD via gluon.Trainer.step, train_yolov3.py:527-530,634)
struct SgdSeg {
  int64_t off, size;      // element range in the flat parameter / gradient / momentum buffers
  float lr_mult, wd_mult;
  int32_t enabled, pad;
};
hipError_t vy_launch_sgd(float* params, const float* grads, float* mom, const SgdSeg* segs_dev,
                         const int32_t* chunk_seg_dev, int n_chunks, float lr, float momentum, float wd,
                         float rescale, hipStream_t s);
#define VY_SGD_CHUNK 4096
