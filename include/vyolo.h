/*
 * vyolo.h — C-ABI of libvyolo.so, the MI355X-native yolo3_darknet53 hot path.
 *
 * The reference has no FFI: its "operator API" for this path is the duck-typed surface of the
 * Gluon HybridBlock that models/definitions/yolo/wrappers.py:9-110 (yolo3_darknet53) returns,
 * as driven by train_yolov3.py and detect_yolo3.py.  Each entry point below names the reference
 * call it stands in for (paths relative to /root/reference).  The Python class that presents
 * the Gluon surface over these entry points is videoyolo_amd/model.py; INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative vy_status and
 *     leaves a message for vy_last_error() (thread-local).
 *   - the library never allocates or frees device memory and never synchronises the device
 *     inside a forward/step call: the caller owns three device buffers per net
 *     (parameters, workspace, and — for training — gradients/momentum), sized by the vy_*_bytes
 *     queries, and passes the HIP stream to launch on (hipStream_t as void*; NULL = default).
 *   - one vy_net per device / rank / stream; a net holds no global state.
 *   - tensors at the boundary use the reference's layouts: images NCHW fp32, conv weights OIHW,
 *     detections (ids (B,post_nms,1), scores (B,post_nms,1), bboxes (B,post_nms,4)) fp32 with
 *     -1 filler, exactly what YOLOV3T.hybrid_forward returns (yolo3.py:1203-1206).
 */
#ifndef VYOLO_H
#define VYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vy_net vy_net;

enum vy_status {
  VY_OK = 0,
  VY_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
  VY_ERR_STATE = -2,        /* call order: buffers not bound, not planned, ... */
  VY_ERR_HIP = -3,          /* a HIP runtime call or kernel launch failed */
  VY_ERR_UNSUPPORTED = -4   /* reference feature outside the hot path (temporal variants ...) */
};

enum vy_param_kind {
  VY_P_WEIGHT = 0, VY_P_GAMMA = 1, VY_P_BETA = 2, VY_P_RUNNING_MEAN = 3, VY_P_RUNNING_VAR = 4,
  VY_P_BIAS = 5
};

/* One row of collect_params() (train_yolov3.py:494-497, wrappers.py:55-57). */
typedef struct vy_param_info {
  char name[96];        /* gluon structural name, e.g. "stages.0.2.body.1.0.weight" */
  int32_t kind;         /* vy_param_kind */
  int32_t ndim;         /* 4 for conv weights (O,I,kh,kw), else 1 */
  int32_t shape[4];     /* reference shape */
  int64_t size;         /* element count */
  int64_t offset;       /* element offset of this tensor inside the device parameter buffer
                           (device layout: conv weights are stored O,kh,kw,I) */
  int32_t trainable;    /* 0 for running_mean / running_var */
  int32_t backbone;     /* 1 if the tensor belongs to the Darknet-53 stages (freeze_base) */
} vy_param_info;

/* Last error message of the calling thread ("" if none). */
const char* vy_last_error(void);

/* Library build id ("vyolo <n> gfx950"). */
const char* vy_version(void);

/* yolo3_darknet53(classes, ...) at k=1 — wrappers.py:9-12,54-58,80-84,101-103 and
 * YOLOV3T.__init__ yolo3.py:959-1054.  num_class = len(classes). */
int vy_net_create(int32_t num_class, vy_net** out);
void vy_net_destroy(vy_net* net);

/* net.set_nms(nms_thresh, nms_topk, post_nms) — yolo3.py:1208-1228.
 *   nms_topk in [1, VY_MAX_TOPK]   the nms_topk best valid candidates go through NMS (the scripts use 400)
 *   nms_topk <= 0                  "-1 to disable": EVERY valid candidate goes through NMS (consumed in
 *                                  score order in chunks of VY_MAX_TOPK until the output rows are filled)
 *   nms_topk > VY_MAX_TOPK         the same chunked kernel, stopped after nms_topk candidates
 *   post_nms > 0                   the outputs have post_nms rows (any value; > VY_MAX_TOPK with a chunked nms_topk: the
 *                                  kept rows are read back from the output instead of living in LDS)
 *   post_nms <= 0                  no slice (yolo3.py:1201-1202): nms_topk rows, or — nms_topk <= 0 too — all N*C rows
 *                                  of box_nms's un-sliced output (vy_net_num_anchors * num_class; quadratic in the worst
 *                                  case, like the reference's loop)
 *   nms_thresh outside (0, 1)      no NMS at all: see vy_net_forward_infer */
#define VY_MAX_TOPK 1024
int vy_net_set_nms(vy_net* net, float nms_thresh, int32_t nms_topk, int32_t post_nms);

/* collect_params(): number of tensors, and row i. */
int32_t vy_net_num_params(const vy_net* net);
int vy_net_param_info(const vy_net* net, int32_t i, vy_param_info* out);

/* The graph the planner executes, one row per convolution in EXECUTION order (host-only; no device needed): what
 * `net.summary(x)` (train_yolov3.py:758) prints per layer, and what tests/test_graph_structure.py compares with the
 * structure the reference's own constructors build (wrappers.py:54-58,80-103; three_darknet.py:162-195;
 * yolo3.py:218-263,1013-1054; layers.py:63-70) — tests/golden/graph_structure.json. */
typedef struct vy_conv_info {
  char name[96];        /* structural prefix of the cell ("stages.0.2.body.1"; the Conv2D itself is "<name>.0") or of the
                           prediction conv ("yolo_outputs.0.prediction") */
  int32_t cin, cout, kernel, stride, pad;
  int32_t has_bn;       /* 1: Conv2D(use_bias=False) -> norm_layer -> LeakyReLU(0.1) (layers.py:63-70); 0: bias, no activation */
  int32_t sync_bn;      /* 1: the cell is built with the norm_layer passed to yolo3_darknet53 (SyncBatchNorm exchanges its
                           statistics there); 0: hard-wired / defaulted BatchNorm (darknet.py:89-91, wrappers.py:101-103) */
  int32_t residual;     /* 1: the block's input is added to this cell's output (darknet.py:40) */
  int32_t upsample;     /* 2: the output is stored x2-replicated and cropped into the concat plane (layers.py:11-20,
                           yolo3.py:1167-1177); 1 otherwise */
  int32_t concat_offset;/* first channel of this conv's output inside its output plane (concat fusion: the upsampled
                           transition comes FIRST, the backbone route after it) */
  int32_t out_channels_total; /* channels of that output plane */
} vy_conv_info;
int32_t vy_net_num_convs(const vy_net* net);
int vy_net_conv_info(const vy_net* net, int32_t i, vy_conv_info* out);

/* Bytes of the device parameter buffer (all tensors + folded-BN scratch). */
size_t vy_net_param_bytes(const vy_net* net);
/* Bind the caller-owned device parameter buffer (net.collect_params().reset_ctx(ctx),
 * detect_yolo3.py:199). */
int vy_net_bind_params(vy_net* net, void* dev_params);

/* net.load_parameters / save_parameters go through these per-tensor copies (train_yolov3.py:
 * 293-303,323-327): host data in the REFERENCE layout; the library converts OIHW <-> device
 * layout.  Both enqueue on `stream` and return after the copy has completed. */
int vy_net_param_set(vy_net* net, int32_t i, const float* host_src, void* stream);
int vy_net_param_get(vy_net* net, int32_t i, float* host_dst, void* stream);

/* Shape planning.  Workspace bytes needed for inference on (batch, 3, height, width) input;
 * height and width in [32, 4096] (alloc_size = (128,128), yolo3.py:67-74).  They need not be multiples of 32:
 * like the reference, every stride-2 conv yields ceil(n / 2) rows and the x2 upsample is cropped to the route it is
 * concatenated with (slice_like, yolo3.py:1177), so the heads have ceil(h / 32), ceil(h / 16), ceil(h / 8) rows. */
size_t vy_net_workspace_bytes(const vy_net* net, int32_t batch, int32_t height, int32_t width);
/* Bind a caller-owned device workspace of at least that size and plan for that shape.  Zeroes the
 * workspace (asynchronously, on `stream`): the padded activation planes rely on zero borders. */
int vy_net_bind_workspace(vy_net* net, void* dev_ws, size_t bytes, int32_t batch, int32_t height,
                          int32_t width, void* stream);

/* Inference activation planes are recycled by liveness (a residual stage needs two alternating block-output planes
 * and one bottleneck plane, not two per block: 608x608 batch 64 plans ~3x less workspace) — the intermediate
 * activations of a finished forward are then gone, like the intermediates of the reference's hybridized graph.
 * keep != 0 gives every cell its own plane so that vy_net_read_activation works (parity taps).  Changes the plan:
 * call before vy_net_workspace_bytes / vy_net_bind_workspace (a bound workspace is unbound by a change).  Training
 * plans always keep every plane (backward reads them). */
int vy_net_set_keep_activations(vy_net* net, int32_t keep);

/* Arithmetic of the inference convolutions (Conv2D inside `_conv2d`, models/definitions/layers.py:63-70; SURVEY 7
 * hard-part (iii) admits "split-fp32").  No counterpart in the reference: mxnet picks its conv algorithm itself.
 *   VY_CONV_EXACT_FP32     default, the parity path: every output one fp32 fma chain on v_mfma_f32_32x32x2_f32,
 *                          bit-identical to oracle/ 
 *   VY_CONV_SPLIT_BF16X3   opt-in, INFERENCE: every conv+BN+leaky cell with cout % 64 == 0 whose launch is large enough
 *                          (per-launch cost model) runs on the bf16 matrix core — every fp32 operand cut exactly into
 *                          three bf16 numbers, six partial products per multiply, fp32 accumulation
 *                          (csrc/conv_split.hip); its long-K 3x3 stride-1 cells in big launches run the same
 *                          arithmetic as a 1-D Winograd F(2, 3) (csrc/conv_wino.hip: a third fewer multiplications).
 *                          NOT bit-equal to the exact path (tolerances: tests/test_gpu_split.py), 1.5x the frames/s at
 *                          608x608 batch 64.  Planes, stem, prediction convs, decode and NMS are shared with the exact
 *                          path.  Training runs the exact kernels.
 *   VY_CONV_SPLIT_BF16X3_TRAIN   as above, and TRAINING too: the recorded forward, the data gradients (conv_split.hip)
 *                          and the weight gradients of every conv with cout % 128 == 0 (wgrad_split.hip) on the bf16
 *                          matrix core; 1.15x the training frames/s at 416x416 batch 16.  Losses within 1e-4 of the exact
 *                          path; gradients as far from it as a one-ulp change of the input moves the exact path's own
 *                          (tests/test_gpu_split.py).
 * Changes the plan (the pre-split weight images live in the workspace): call before vy_net_workspace_bytes /
 * vy_net_bind_workspace; a bound workspace is unbound by a change. */
enum vy_conv_mode { VY_CONV_EXACT_FP32 = 0, VY_CONV_SPLIT_BF16X3 = 1, VY_CONV_SPLIT_BF16X3_TRAIN = 2 };
int vy_net_set_conv_mode(vy_net* net, int32_t mode);
int32_t vy_net_get_conv_mode(const vy_net* net);
/* The weight images of VY_CONV_SPLIT_BF16X3 are rebuilt by the next inference forward after any parameter write the
 * library performs itself (vy_net_bind_params, vy_net_param_set, vy_net_sgd_step, vy_net_bind_workspace).  A caller
 * that writes the device parameter buffer directly (the Trainer's broadcast from rank 0, train_yolov3.py:527-530)
 * says so with this call. */
int vy_net_invalidate_split_weights(vy_net* net);

/* Diagnostics of the chain-preserving stream-K conv launches (csrc/conv_igemm.hip; no counterpart in the reference).
 * enabled: 1 if the bind-time probe saw the workgroup placement the schedule is built for (MI355X, SPX mode: 8 XCDs,
 * blocks L and L + 8 on one XCD, 256 CUs) — otherwise every conv is a plain launch.  flags_offset / n_flags: where the
 * hand-off flags live in the bound workspace (32-bit words, all zero between launches; after ANY entry point of the
 * handle has returned an error the next forward / training step zeroes them before it launches).  Tests only. */
int vy_net_streamk_state(const vy_net* net, int32_t* enabled, size_t* flags_offset, int32_t* n_flags);

/* Number of anchors N = 3 * sum_i (H/s_i)(W/s_i) for the planned shape. */
int32_t vy_net_num_anchors(const vy_net* net);

/* net(x) outside autograd — YOLOV3T.hybrid_forward inference branch, yolo3.py:1076-1206:
 * Darknet-53 stages -> 3 detection blocks/outputs -> decode -> box_nms -> first post_nms rows.
 *   x        device, (batch,3,H,W) fp32 NCHW
 *   ids      device, (batch,post_nms,1)     scores  device, (batch,post_nms,1)
 *   bboxes   device, (batch,post_nms,4)     corner format, input-pixel units, un-clipped
 *   keep_idx device, (batch,post_nms) int32, nullable: row index into the reference's
 *            pre-NMS (B, N*C, 6) detection tensor for every returned row (-1 for filler).
 * With nms_thresh outside (0,1) the reference skips box_nms and the slice (yolo3.py:1197-1202): the
 * outputs then have vy_net_num_anchors()*num_class rows — the (B, N*C, 6) detection tensor itself in its
 * class-major row order — and keep_idx is the row number.
 * Asynchronous on `stream`. */
int vy_net_forward_infer(vy_net* net, const float* x, float* ids, float* scores, float* bboxes,
                         int32_t* keep_idx, void* stream);

/* The detection tail ALONE, on caller-supplied prediction-conv outputs: YOLOOutputV3.hybrid_forward's inference branch
 * (yolo3.py:158-197: decode, x C tile, class-major rows) for the three scales, their concat (yolo3.py:1195), box_nms and
 * the slice (yolo3.py:1197-1206) — what `net.yolo_outputs[i](pred)` + `F.contrib.box_nms` compute in the reference.
 *   head_i   device, (batch, 3*(5+C), H_i, W_i) fp32 NCHW for strides 32, 16, 8 (the layout vy_net_read_head returns),
 *            H_i x W_i of the bound workspace's plan
 * Outputs as vy_net_forward_infer (nms_thresh outside (0,1): the (B, N*C, 6) detection tensor itself).  This is the
 * operator-level door the golden-vector kit uses (tests/golden/make_mxnet_goldens.py: hand-built logits decide threshold
 * strictness, tie order and the top-k cut on the SAME kernels a forward runs); it overwrites the head planes of the
 * workspace.  Asynchronous on `stream`. */
int vy_net_detect_heads(vy_net* net, const float* head0, const float* head1, const float* head2, float* ids,
                        float* scores, float* bboxes, int32_t* keep_idx, void* stream);

/* Debug / parity taps (asynchronous on `stream`, valid after a forward on the same stream):
 * copy head i's prediction-conv output (yolo3.py:154 `pred`) to dst as (batch, 3*(5+C), H_i, W_i)
 * NCHW — i = 0,1,2 for strides 32,16,8. */
int vy_net_read_head(vy_net* net, int32_t i, float* dst_dev, void* stream);
/* copy the activation of feature cell `name` ("stages.0.14", "yolo_blocks.1.tip", ...) to dst
 * as NCHW; returns its channel count / height / width through the out pointers.  With dst != NULL it needs a plan
 * that keeps every plane (vy_net_set_keep_activations, or a training plan): VY_ERR_STATE otherwise. */
int vy_net_read_activation(vy_net* net, const char* name, float* dst_dev, int32_t* c, int32_t* h,
                           int32_t* w, void* stream);

/* Per-launch device timing of the last forward: runs one forward with HIP events recorded on
 * `stream` around every kernel launch and returns, for launch j < *n, its name, the kernel
 * time in ms and its algorithmic FLOPs (2*MAC; 0 for non-conv launches).  Synchronises. */
typedef struct vy_launch_stat {
  char name[64];
  float ms;
  double flops;
  double bytes;   /* algorithmic HBM bytes (inputs read once + outputs written once) */
} vy_launch_stat;
int vy_net_profile_infer(vy_net* net, const float* x, float* ids, float* scores, float* bboxes,
                         vy_launch_stat* stats, int32_t* n, void* stream);

/* A HIP stream owned by the caller and created by the library (hipStreamCreateWithFlags, non-blocking):
 * its own hardware queue, for callers that run two launch sequences side by side — the reference's
 * `for x in data:` loop over per-device batches (detect_yolo3.py:211-222) on ONE device's two half
 * batches.  (torch's pooled streams may share the default stream's queue.) */
int vy_stream_create(void** stream);
int vy_stream_destroy(void* stream);

/* Frame pre-processing in front of the path: (batch,H,W,3) uint8 HWC device
 * frames -> (batch,3,H,W) fp32 NCHW, y = (x/255 - mean[c]) / std[c] — mx.nd.image.to_tensor +
 * mx.nd.image.normalize at models/definitions/yolo/transforms.py:331-334.  mean3/std3 are host
 * pointers to 3 floats.  Frames that are not at the network size yet: vy_preprocess_resize_frames below. */
int vy_preprocess_frames(const uint8_t* frames_hwc, float* out_nchw, int32_t batch, int32_t height,
                         int32_t width, const float* mean3, const float* std3, void* stream);

/* The whole YOLO3VideoInferenceTransform.__call__ (models/definitions/yolo/transforms.py:316-350) in one
 * launch: (batch, src_height, src_width, 3) uint8 device frames -> resize to (height, width) as
 * timage.imresize(frame, width, height, interp=9) does (:325-327: OpenCV INTER_AREA when both sides shrink,
 * INTER_CUBIC when both grow, INTER_LINEAR otherwise, on uint8 with OpenCV's fixed-point / float arithmetic
 * [UPSTREAM-RECALLED: gluoncv, mxnet and OpenCV are not available here; oracle/resize_oracle.py states the
 * arithmetic and tests/test_resize_oracle.py cross-checks it against torch / exact area definitions]) -> the
 * uint8 value the reference's resized NDArray would hold -> to_tensor + normalize -> (batch,3,height,width)
 * fp32 NCHW.  The resized frame itself is never materialised.  Area shrink factors above 10 are rejected. */
int vy_preprocess_resize_frames(const uint8_t* frames_hwc, int32_t src_height, int32_t src_width, float* out_nchw,
                                int32_t batch, int32_t height, int32_t width, const float* mean3, const float* std3,
                                void* stream);

/* Prefetch target generation on the device: YOLOV3PrefetchTargetGenerator.forward,
 * models/definitions/yolo/yolo_target.py:31-148 (called per sample from the DataLoader transform,
 * transforms.py:259-277), for a whole batch.  gt_boxes (batch,num_gt,4) corner pixels of the
 * (height,width) network input, gt_ids (batch,num_gt) class index as fp32, gt_mixratio (batch,num_gt)
 * or NULL; a row with any negative coordinate ends that image's list (:107-108).  Outputs, all fp32
 * device buffers fully written: objness_t (batch,N,1), centers_t / scales_t / weights_t (batch,N,2),
 * clas_t (batch,N,num_class), N = vy_net_num_anchors order (stride 32,16,8 -> cell -> anchor) — exactly
 * the five tensors vy_net_train_forward takes.  Anchors are the yolo3_darknet53 table (wrappers.py:80-84). */
int vy_prefetch_targets(const float* gt_boxes, const float* gt_ids, const float* gt_mixratio, int32_t batch,
                        int32_t num_gt, int32_t height, int32_t width, int32_t num_class, float* objness_t,
                        float* centers_t, float* scales_t, float* weights_t, float* clas_t, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training step.  Reference call pattern, train_yolov3.py:623-634:
 *     with autograd.record():
 *         obj, ctr, scl, cls = net(x, gt_boxes, obj_t, centers_t, scales_t, weights_t, clas_t)
 *         autograd.backward(obj + ctr + scl + cls)
 *     trainer.step(batch_size)
 * The caller owns two more device buffers of vy_net_param_bytes each (gradients, SGD momentum;
 * same element offsets as the parameter buffer: one contiguous range for the RCCL all-reduce) and a
 * training workspace (activations, raw conv outputs / their gradients, gradient planes, scratch).
 * ---------------------------------------------------------------------------------------------- */
/* Training shapes: height and width multiples of 32, as train_yolov3.py produces them (0 / VY_ERR_UNSUPPORTED
 * otherwise). */
size_t vy_net_train_workspace_bytes(const vy_net* net, int32_t batch, int32_t height, int32_t width);
/* Binds the training workspace (also serves inference at that shape), the gradient buffer and the
 * momentum buffer (the caller zero-initialises the momentum once); zeroes the workspace
 * asynchronously on `stream`. */
int vy_net_bind_train(vy_net* net, void* dev_ws, size_t bytes, int32_t batch, int32_t height,
                      int32_t width, void* dev_grads, void* dev_momentum, void* stream);

/* YOLOV3T(ignore_iou_thresh=0.7) (yolo3.py:962) and net._target_generator._label_smooth
 * (train_yolov3.py:499-500, yolo_target.py:272-278). */
int vy_net_set_train_options(vy_net* net, float ignore_iou_thresh, int32_t label_smooth);

/* net(x, gt_boxes, *fixed_targets) under autograd.record() — yolo3.py:1126-1187 with BatchNorm on
 * batch statistics (running stats updated, momentum 0.9), YOLOV3TargetMerger (yolo_target.py:
 * 226-281) and YOLOV3Loss.  Also computes d(sum of the four losses)/d(raw predictions), so that
 * vy_net_train_backward only has to walk the network.
 *   x (B,3,H,W)  gt_boxes (B,M,4) corner px, -1 padded   obj_t (B,N,1)  centers_t/scales_t/weights_t
 *   (B,N,2)  clas_t (B,N,C) — N anchors in the reference order (stride 32,16,8; cell; anchor)
 *   losses: device (4,B): obj, center, scale, cls per sample. */
int vy_net_train_forward(vy_net* net, const float* x, const float* gt_boxes, int32_t M,
                         const float* obj_t, const float* centers_t, const float* scales_t,
                         const float* weights_t, const float* clas_t, float* losses, void* stream);

/* net(x) under autograd.train_mode() without recording — yolo3.py:1189-1192, the branch the DataLoader
 * transform drives (transforms.py:190-193): the network runs with BatchNorm on batch statistics (running
 * stats updated, as mxnet's BatchNorm does whenever is_training), and the per-anchor tensors of
 * YOLOOutputV3's training return (yolo3.py:179-182) come back concatenated over the scales (stride 32, 16,
 * 8 -> cell -> anchor):  box_preds (B,N,4) decoded corner boxes, centers (B,N,2) / scales (B,N,2) /
 * objness (B,N,1) / class_pred (B,N,C) RAW predictions.  Items 1-3 of the reference's 8-tuple (anchors,
 * offsets, fake feature maps) are constants of the input shape and are built by the host mirror.
 * Needs the training workspace (vy_net_bind_train).  All outputs are device buffers. */
int vy_net_train_mode_forward(vy_net* net, const float* x, float* box_preds, float* centers, float* scales,
                              float* objness, float* class_pred, void* stream);

/* autograd.backward(sum_losses) (train_yolov3.py:631): fills the gradient buffer (every trainable
 * tensor, device layout) from the state left by the last vy_net_train_forward.  `x` is the same
 * image batch (needed by the stem's weight gradient). */
int vy_net_train_backward(vy_net* net, const float* x, void* stream);

/* Per-parameter optimizer attributes: Parameter.lr_mult / wd_mult (train_yolov3.py:496-497) and
 * grad_req = 'null' (enabled = 0; wrappers.py:55-57 freeze_base). */
int vy_net_param_set_opt(vy_net* net, int32_t i, float lr_mult, float wd_mult, int32_t enabled);

/* trainer.step(batch_size) for Trainer('sgd', {wd, momentum}) (train_yolov3.py:527-530,634):
 * g = rescale_grad*grad + wd*w ; mom = momentum*mom - lr*g ; w += mom.  rescale_grad = 1/batch_size.
 * Gradients must already be summed across ranks (all-reduce of the gradient buffer). */
int vy_net_sgd_step(vy_net* net, float lr, float momentum, float wd, float rescale_grad, void* stream);

/* Gradient of parameter i in the REFERENCE layout to host memory (tests / checkpoints). */
int vy_net_grad_get(vy_net* net, int32_t i, float* host_dst, void* stream);

/* Parity tap: gradient w.r.t. the output of cell `name` after vy_net_train_backward, as NCHW at the
 * resolution the output is stored (x2 for the transition cells). */
int vy_net_read_grad_activation(vy_net* net, const char* name, float* dst_dev, void* stream);

/* SyncBatchNorm(num_devices) (train_yolov3.py:352-354).  With world > 1 the BatchNorm layers that
 * the reference builds with the passed norm_layer — the stem and the five stride-2 convs of
 * Darknet-53 (three_darknet.py:163-181; the residual blocks hard-code BatchNorm, :193-194, and
 * wrappers.py:101-103 does not forward norm_layer to YOLOV3T) — call `cb` to sum their [2][C]
 * double-precision statistics over all ranks, forward and backward.  cb(user, device_ptr, count)
 * must all-reduce (sum) `count` doubles in place, ordered with the stream passed to the step. */
typedef int (*vy_allreduce_cb)(void* user, void* dev_ptr, int64_t count);
int vy_net_set_sync_bn(vy_net* net, int32_t world, vy_allreduce_cb cb, void* user);

/* Bucketed gradient exchange: during vy_net_train_backward `cb(user, elem_offset, elem_count)` is
 * called each time a contiguous range of the gradient buffer is final (heads first, then Darknet
 * stages 2, 1, 0), after the kernels producing it were enqueued — the caller records an event and
 * all-reduces that range on a side stream, overlapping the rest of the backward pass. */
typedef int (*vy_grad_bucket_cb)(void* user, int64_t elem_offset, int64_t elem_count);
int vy_net_set_grad_bucket_cb(vy_net* net, vy_grad_bucket_cb cb, void* user);

#ifdef __cplusplus
}
#endif
#endif /* VYOLO_H */
