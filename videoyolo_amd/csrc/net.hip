// net.hip — host side of libvyolo.so: the yolo3_darknet53 graph, parameter table, activation
// planner and the C-ABI of include/vyolo.h.
//
// The graph is the reference's (paths relative to /root/reference):
//   Darknet-53 stages       models/definitions/darknet/three_darknet.py:162-195, 252-258
//                           sliced features[:15] / [15:24] / [24:]  (yolo/wrappers.py:58)
//   detection blocks/heads  models/definitions/yolo/yolo3.py:218-263, 1013-1054
//   forward order           models/definitions/yolo/yolo3.py:1105-1206
// but it is executed as a flat list of fused launches over zero-bordered NHWC planes (kernels.h):
// conv+BN+leaky(+residual) is one kernel, upsample+concat costs nothing (the transition conv
// stores x2-replicated straight into the channel range [0,c) of the concat plane, whose range
// [c, ..) the backbone stage wrote in place), decode+NMS never materialises (B, N*C, 6).
#include "net_internal.h"

static thread_local std::string g_err;

#undef fail
int vy_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define fail vy_fail

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* vy_last_error(void) { return g_err.c_str(); }
const char* vy_version(void) { return "vyolo 1 gfx950"; }

int vy_net_create(int32_t num_class, vy_net** out) {
  if (!out) return fail(VY_ERR_INVALID, "out is null");
  if (num_class < 1 || num_class > 1000) return fail(VY_ERR_INVALID, "num_class %d out of range", num_class);
  vy_net* n = new vy_net();
  n->num_class = num_class;
  n->build();
  *out = n;
  return 0;
}

void vy_net_destroy(vy_net* net) {
  if (!net) return;
  vy_train_free(net);
  delete net;
}

int vy_net_set_nms(vy_net* net, float nms_thresh, int32_t nms_topk, int32_t post_nms) {
  if (!net) return fail(VY_ERR_INVALID, "net is null");
  net->nms_thresh = nms_thresh;
  net->nms_topk = nms_topk;
  net->post_nms = post_nms;
  return 0;
}

int32_t vy_net_num_params(const vy_net* net) { return net ? (int32_t)net->params.size() : 0; }

int vy_net_param_info(const vy_net* net, int32_t i, vy_param_info* out) {
  if (!net || !out || i < 0 || i >= (int32_t)net->params.size()) return fail(VY_ERR_INVALID, "bad param index %d", i);
  *out = net->params[i].info;
  return 0;
}

int32_t vy_net_num_convs(const vy_net* net) { return net ? (int32_t)net->convs.size() : 0; }

int vy_net_conv_info(const vy_net* net, int32_t i, vy_conv_info* out) {
  if (!net || !out || i < 0 || i >= (int32_t)net->convs.size()) return fail(VY_ERR_INVALID, "bad conv index %d", i);
  const ConvT& c = net->convs[i];
  memset(out, 0, sizeof *out);
  snprintf(out->name, sizeof out->name, "%s", c.name.c_str());
  out->cin = c.cin;
  out->cout = c.cout;
  out->kernel = c.k;
  out->stride = c.stride;
  out->pad = c.k / 2;
  out->has_bn = c.p_gamma >= 0;
  out->sync_bn = is_sync_layer(c);
  out->residual = c.res_plane >= 0;
  out->upsample = c.ups;
  out->concat_offset = c.out_co;
  out->out_channels_total = net->planes[c.out_plane].C;
  return 0;
}

size_t vy_net_param_bytes(const vy_net* net) { return net ? (size_t)net->param_elems * sizeof(float) : 0; }

int vy_net_bind_params(vy_net* net, void* dev_params) {
  if (!net || !dev_params) return fail(VY_ERR_INVALID, "null argument");
  net->dev_params = static_cast<float*>(dev_params);
  net->split_dirty = net->dsplit_dirty = net->wino_dirty = true;
  return 0;
}

// OIHW (reference) <-> O,kh,kw,I (device)
static void pack_oihw(const float* src, float* dst, int O, int I, int k) {
  for (int o = 0; o < O; ++o)
    for (int i = 0; i < I; ++i)
      for (int t = 0; t < k * k; ++t) dst[((size_t)o * k * k + t) * I + i] = src[((size_t)o * I + i) * k * k + t];
}
static void unpack_oihw(const float* src, float* dst, int O, int I, int k) {
  for (int o = 0; o < O; ++o)
    for (int i = 0; i < I; ++i)
      for (int t = 0; t < k * k; ++t) dst[((size_t)o * I + i) * k * k + t] = src[((size_t)o * k * k + t) * I + i];
}

int vy_net_param_set(vy_net* net, int32_t i, const float* host_src, void* stream) {
  if (!net || !host_src || i < 0 || i >= (int32_t)net->params.size()) return fail(VY_ERR_INVALID, "bad argument");
  if (!net->dev_params) return fail(VY_ERR_STATE, "parameters not bound");
  const vy_param_info& pi = net->params[i].info;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* dst = net->dev_params + pi.offset;
  net->split_dirty = net->dsplit_dirty = net->wino_dirty = true;
  if (pi.ndim == 4) {
    std::vector<float> tmp((size_t)pi.size);
    pack_oihw(host_src, tmp.data(), pi.shape[0], pi.shape[1], pi.shape[2]);
    HIP_TRY(hipMemcpyAsync(dst, tmp.data(), sizeof(float) * pi.size, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
  } else {
    HIP_TRY(hipMemcpyAsync(dst, host_src, sizeof(float) * pi.size, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return 0;
}

int vy_net_param_get(vy_net* net, int32_t i, float* host_dst, void* stream) {
  if (!net || !host_dst || i < 0 || i >= (int32_t)net->params.size()) return fail(VY_ERR_INVALID, "bad argument");
  if (!net->dev_params) return fail(VY_ERR_STATE, "parameters not bound");
  const vy_param_info& pi = net->params[i].info;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* src = net->dev_params + pi.offset;
  if (pi.ndim == 4) {
    std::vector<float> tmp((size_t)pi.size);
    HIP_TRY(hipMemcpyAsync(tmp.data(), src, sizeof(float) * pi.size, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    unpack_oihw(tmp.data(), host_dst, pi.shape[0], pi.shape[1], pi.shape[2]);
  } else {
    HIP_TRY(hipMemcpyAsync(host_dst, src, sizeof(float) * pi.size, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  return 0;
}

static int check_shape(int32_t batch, int32_t h, int32_t w) {
  if (batch < 1) return fail(VY_ERR_INVALID, "batch %d < 1", batch);
  if (h < 32 || w < 32 || h > 4096 || w > 4096)
    return fail(VY_ERR_INVALID, "input %dx%d: height and width must lie in [32, 4096]", h, w);
  return 0;
}

size_t vy_net_workspace_bytes(const vy_net* net, int32_t batch, int32_t height, int32_t width) {
  if (!net || check_shape(batch, height, width)) return 0;
  const_cast<vy_net*>(net)->resolve_cus();
  return const_cast<vy_net*>(net)->plan(batch, height, width, false, net->keep_activations);
}

int vy_net_bind_workspace(vy_net* net, void* dev_ws, size_t bytes, int32_t batch, int32_t height, int32_t width,
                          void* stream) {
  if (!net || !dev_ws) return fail(VY_ERR_INVALID, "null argument");
  if (int rc = check_shape(batch, height, width)) return rc;
  if (int rc = net->bind_cus(dev_ws)) return rc;
  const size_t need = net->plan(batch, height, width, false, net->keep_activations);
  if (bytes < need) return fail(VY_ERR_INVALID, "workspace too small: %zu < %zu bytes", bytes, need);
  net->plan(batch, height, width, true, net->keep_activations);
  net->dev_ws = static_cast<unsigned char*>(dev_ws);
  net->ws_bytes = bytes;
  net->fold_uploaded = false;
  HIP_TRY(hipMemsetAsync(dev_ws, 0, need, static_cast<hipStream_t>(stream)));
  net->sk_dirty = false;
  net->sk_ok = vy_sk_verify_topology(reinterpret_cast<unsigned*>(net->dev_ws + net->sk_off), static_cast<hipStream_t>(stream)) != 0;
  return 0;
}

int vy_net_set_keep_activations(vy_net* net, int32_t keep) {
  if (!net) return fail(VY_ERR_INVALID, "net is null");
  if ((keep != 0) != net->keep_activations) {
    net->keep_activations = keep != 0;
    net->dev_ws = nullptr;  // the plan changed: the workspace has to be sized and bound again
    net->ws_bytes = 0;
  }
  return 0;
}

int vy_net_set_conv_mode(vy_net* net, int32_t mode) {
  if (!net) return fail(VY_ERR_INVALID, "net is null");
  if (mode < VY_CONV_EXACT_FP32 || mode > VY_CONV_SPLIT_BF16X3_TRAIN) return fail(VY_ERR_INVALID, "conv mode %d", mode);
  if (mode != net->conv_mode) {
    net->conv_mode = mode;
    net->dev_ws = nullptr;  // the plan changed (the weight images live in the workspace): size and bind again
    net->ws_bytes = 0;
  }
  return 0;
}

int32_t vy_net_get_conv_mode(const vy_net* net) { return net ? net->conv_mode : 0; }

int vy_net_streamk_state(const vy_net* net, int32_t* enabled, size_t* flags_offset, int32_t* n_flags) {
  if (!net) return fail(VY_ERR_INVALID, "net is null");
  if (!net->dev_ws) return fail(VY_ERR_STATE, "workspace not bound");
  if (enabled) *enabled = net->sk_ok ? 1 : 0;
  if (flags_offset) *flags_offset = net->sk_off;
  if (n_flags) *n_flags = VY_SK_FLAGS;
  return 0;
}

int vy_net_invalidate_split_weights(vy_net* net) {
  if (!net) return fail(VY_ERR_INVALID, "net is null");
  net->split_dirty = net->dsplit_dirty = net->wino_dirty = true;
  return 0;
}

int32_t vy_net_num_anchors(const vy_net* net) {
  if (!net || !net->dev_ws) return 0;
  int n = 0;
  for (int i = 0; i < 3; ++i) n += 3 * net->planes[net->head_plane[i]].H * net->planes[net->head_plane[i]].W;
  return n;
}

int vy_net_forward_infer(vy_net* net, const float* x, float* ids, float* scores, float* bboxes, int32_t* keep_idx,
                         void* stream) {
  if (net && (!x || !ids || !scores || !bboxes)) net->sk_dirty = true;  // (any error return: see vy_net::sk_dirty)
  if (!net || !x || !ids || !scores || !bboxes) return fail(VY_ERR_INVALID, "null argument");
  if (int rc = net->sk_begin(static_cast<hipStream_t>(stream))) return rc;
  return net->sk_end(net->forward<false>(x, ids, scores, bboxes, keep_idx, static_cast<hipStream_t>(stream),
                                         [](const char*, double, double, bool) {}));
}

int vy_net_read_head(vy_net* net, int32_t i, float* dst_dev, void* stream) {
  if (!net || !dst_dev || i < 0 || i > 2) return fail(VY_ERR_INVALID, "bad argument");
  if (int rc = net->check_ready()) return rc;
  const PlaneT& p = net->planes[net->head_plane[i]];
  HIP_TRY(vy_launch_plane_to_nchw(net->plane_ptr(net->head_plane[i]), net->B, p.H, p.W, p.C, 0,
                                  3 * (5 + net->num_class), dst_dev, static_cast<hipStream_t>(stream)));
  return 0;
}

int vy_net_detect_heads(vy_net* net, const float* head0, const float* head1, const float* head2, float* ids, float* scores,
                        float* bboxes, int32_t* keep_idx, void* stream) {
  if (!net || !head0 || !head1 || !head2 || !ids || !scores || !bboxes) return fail(VY_ERR_INVALID, "null argument");
  if (int rc = net->check_ready()) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* src[3] = {head0, head1, head2};
  for (int i = 0; i < 3; ++i) {
    const PlaneT& p = net->planes[net->head_plane[i]];
    HIP_TRY(vy_launch_nchw_to_plane(src[i], net->B, p.H, p.W, p.C, 0, 3 * (5 + net->num_class),
                                    net->plane_ptr(net->head_plane[i]), s));
  }
  const DetArgs d = net->det_args();
  if (net->nms_thresh > 0.f && net->nms_thresh < 1.f)  // yolo3.py:1197
    HIP_TRY(vy_launch_detect(d, net->dev_ws + net->det_scratch_off, ids, scores, bboxes, keep_idx, s));
  else
    HIP_TRY(vy_launch_raw_detections(d, ids, scores, bboxes, keep_idx, s));
  return 0;
}

int vy_net_read_activation(vy_net* net, const char* name, float* dst_dev, int32_t* c, int32_t* h, int32_t* w,
                           void* stream) {
  if (!net || !name) return fail(VY_ERR_INVALID, "bad argument");
  if (int rc = net->check_ready()) return rc;
  if (dst_dev && net->planes_shared)
    return fail(VY_ERR_STATE, "activation planes are recycled in this plan: call vy_net_set_keep_activations(net, 1) "
                "before sizing / binding the workspace to read intermediate activations");
  for (const ConvT& cv : net->convs) {
    if (cv.name != name) continue;
    const PlaneT& p = net->planes[cv.out_plane];
    if (c) *c = cv.cout;
    if (h) *h = p.H;
    if (w) *w = p.W;
    if (dst_dev)
      HIP_TRY(vy_launch_plane_to_nchw(net->plane_ptr(cv.out_plane), net->B, p.H, p.W, p.C, cv.out_co, cv.cout,
                                      dst_dev, static_cast<hipStream_t>(stream)));
    return 0;
  }
  return fail(VY_ERR_INVALID, "no cell named '%s'", name);
}

int vy_net_profile_infer(vy_net* net, const float* x, float* ids, float* scores, float* bboxes,
                         vy_launch_stat* stats, int32_t* n, void* stream) {
  if (!net || !stats || !n) return fail(VY_ERR_INVALID, "null argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int cap = *n;
  std::vector<hipEvent_t> ev0, ev1;
  std::vector<vy_launch_stat> rec;
  bool bad = false;
  auto hook = [&](const char* name, double fl, double by, bool before) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {
      bad = true;
      return;
    }
    (void)hipEventRecord(e, s);
    if (before) {
      vy_launch_stat st;
      memset(&st, 0, sizeof st);
      snprintf(st.name, sizeof st.name, "%s", name);
      st.flops = fl;
      st.bytes = by;
      rec.push_back(st);
      ev0.push_back(e);
    } else {
      ev1.push_back(e);
    }
  };
  int rc = net->sk_begin(s);
  if (rc == 0) rc = net->sk_end(net->forward(x, ids, scores, bboxes, nullptr, s, hook));
  if (rc == 0) {
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = fail(VY_ERR_HIP, "sync: %s", hipGetErrorString(e));
  }
  int cnt = 0;
  for (size_t i = 0; i < ev1.size() && i < ev0.size(); ++i) {
    if (rc == 0) (void)hipEventElapsedTime(&rec[i].ms, ev0[i], ev1[i]);
    if ((int)i < cap) stats[i] = rec[i], cnt = (int)i + 1;
  }
  for (auto e : ev0) (void)hipEventDestroy(e);
  for (auto e : ev1) (void)hipEventDestroy(e);
  *n = cnt;
  if (bad && rc == 0) rc = fail(VY_ERR_HIP, "hipEventCreate failed");
  return rc;
}

int vy_stream_create(void** stream) {
  if (!stream) return fail(VY_ERR_INVALID, "null argument");
  hipStream_t s = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (e != hipSuccess) return fail(VY_ERR_HIP, "hipStreamCreateWithFlags: %s", hipGetErrorString(e));
  *stream = s;
  return 0;
}

int vy_stream_destroy(void* stream) {
  if (!stream) return 0;
  hipError_t e = hipStreamDestroy(static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(VY_ERR_HIP, "hipStreamDestroy: %s", hipGetErrorString(e));
  return 0;
}

}  // extern "C"
