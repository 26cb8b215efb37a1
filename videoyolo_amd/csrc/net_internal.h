// net_internal.h — the vy_net object shared by net.hip (graph, planner, inference C-ABI) and
// train.hip (training planner and C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vyolo.h"
#include "kernels.h"
#include "../../include/vy_math.h"

// error channel (defined in net.hip)
int vy_fail(int code, const char* fmt, ...);
#define fail vy_fail

#define HIP_TRY(expr)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) return fail(VY_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct ParamT {
  vy_param_info info;
};

struct PlaneT {
  int C = 0;         // channel stride
  int div = 1;       // spatial = input / div
  size_t off = 0;    // float offset in workspace (set by plan)
  int H = 0, W = 0;  // set by plan
};

struct ConvT {
  std::string name;      // structural prefix of the cell, e.g. "stages.0.2.body.1"
  int in_plane, in_co, cin;
  int out_plane, out_co, cout;
  int k, stride, leaky, ups;
  int res_plane, res_co;          // -1: none
  int p_weight;                   // param indices
  int p_gamma, p_beta, p_mean, p_var;  // -1 for the prediction convs
  int p_bias;                     // -1 for cells
  int64_t scale_off, shift_off;   // folded BN scratch (element offsets), -1 if none
  int is_stem;
  int64_t wino_off = -1;          // ... of its Winograd-transformed image sets (conv_wino.hip: 3x3 stride-1 cells with cout % 128 == 0), -1: none
  int64_t split_off = -1;         // conv mode VY_CONV_SPLIT_BF16X3: byte offset of the bf16 weight images in the workspace region, -1: exact kernel
};

// the cells Darknet3D builds with the norm_layer passed to yolo3_darknet53 — the stem and the five stride-2 convs
// (three_darknet.py:163-181); every other BatchNorm of the model is a plain per-device one: the residual blocks hard-wire
// BatchNorm (darknet.py:89-91) and wrappers.py:101-103 does not hand norm_layer to YOLOV3T, so the heads default to it
inline bool is_sync_layer(const ConvT& c) {
  return c.p_gamma >= 0 && c.name.rfind("stages.", 0) == 0 && c.name.find(".body.") == std::string::npos;
}

static const int kAnchors[3][6] = {{10, 13, 16, 30, 33, 23}, {30, 61, 62, 45, 59, 119}, {116, 90, 156, 198, 373, 326}};
static const int kStrides[3] = {8, 16, 32};  // wrappers.py:80-84

struct VyTrain;
void vy_train_free(struct vy_net* net);

struct vy_net {
  int num_class = 0;
  float nms_thresh = 0.45f;  // YOLOV3T.__init__ defaults, yolo3.py:959-963
  int nms_topk = 400, post_nms = 100;
  std::vector<ParamT> params;
  std::vector<PlaneT> planes;
  std::vector<ConvT> convs;
  int head_plane[3] = {-1, -1, -1};  // prediction conv outputs, order stride 32,16,8
  int64_t param_elems = 0;           // tensors + scratch
  int64_t tensor_elems = 0;
  std::vector<FoldDesc> folds;
  // bound state
  float* dev_params = nullptr;
  unsigned char* dev_ws = nullptr;
  size_t ws_bytes = 0;
  int B = 0, H = 0, W = 0;
  size_t fold_desc_off = 0, det_scratch_off = 0, planes_off = 0, sk_off = 0;  // byte offsets in workspace
  size_t ck_off = 0, ck_bytes = 0;  // running-sum scratch of the convs whose K is summed in runs (conv_igemm.hip)
  bool fold_uploaded = false;
  bool keep_activations = false;  // vy_net_set_keep_activations: inference planes are not recycled (parity taps)
  bool planes_shared = false;     // the committed plan recycles planes (read_activation is then meaningless)
  // vy_net_set_conv_mode: VY_CONV_SPLIT_BF16X3 sends the inference launches conv_split.hip can serve through the bf16
  // matrix core (bf16 x 3, six products, fp32 accumulate); their pre-split weight images live in the workspace and are
  // rebuilt by the next forward whenever the parameters may have changed (split_dirty)
  // stream-K bookkeeping: sk_ok = the device's workgroup placement was verified at bind time (vy_sk_verify_topology);
  // sk_dirty = an entry point of this handle returned an error after it may have launched stream-K kernels: the hand-off
  // flags are cleared by their consumers, so an aborted sequence can leave one up — the next forward / step zeroes them
  bool sk_ok = false;
  bool sk_dirty = false;
  int conv_mode = 0;
  bool split_dirty = true;    // the forward weight images are stale
  bool dsplit_dirty = true;   // the data-gradient weight images (training plans; train.hip) are stale
  // the Winograd image sets (conv_wino.hip) are stale.  Their own flag: a training plan holds them too (model.py reuses it
  // for inference at the same size) but train.hip's refresh_split_images rebuilds only the forward and data-gradient sets —
  // only forward() below, which does rebuild them, may clear it
  bool wino_dirty = true;
  // VY_SPLIT_ALWAYS / VY_SPLIT_WINO (test switches of the per-launch choice), read once at the start of every forward /
  // training step (refresh_env) instead of by every launch's vy_conv_*_pays; -1: not read yet
  int env_split_always = -1, env_wino_mode = -1;
  // CUs of THIS net's device: resolved once, by the first sizing / bind call, from the calling thread's current device —
  // and checked at every bind against the device that owns the workspace (bind_cus below), so that a C-ABI caller whose
  // current device differs between vy_net_*workspace_bytes and vy_net_bind_* gets an error instead of a plan sized for
  // one chip and bound on another.  Every cost-model choice reads this through ConvArgs::cus.
  int cus = 0;
  int resolve_cus() {
    if (!cus) cus = vy_cu_count();
    return cus;
  }
  int bind_cus(const void* ws) {
    const int sized = resolve_cus();
    const int owner = vy_cu_count_of_ptr(ws);
    if (owner && owner != sized)
      return fail(VY_ERR_STATE,
                  "the workspace lives on a device with %d CUs but this net sized its plan for %d (the calling thread's current "
                  "device at the first vy_net_*workspace_bytes call): size and bind with the same current device", owner, sized);
    return 0;
  }
  void refresh_env() {
    const char* a = getenv("VY_SPLIT_ALWAYS");
    const char* w = getenv("VY_SPLIT_WINO");
    env_split_always = a && atoi(a) ? 1 : 0;
    env_wino_mode = w ? atoi(w) : 1;
  }
  size_t wsplit_off = 0;
  struct VyTrain* train = nullptr;  // training planner state, owned by train.hip

  int add_param(const std::string& name, int kind, int ndim, const int* shape, int trainable, int backbone) {
    ParamT p;
    memset(&p.info, 0, sizeof p.info);
    snprintf(p.info.name, sizeof p.info.name, "%s", name.c_str());
    p.info.kind = kind;
    p.info.ndim = ndim;
    int64_t sz = 1;
    for (int i = 0; i < ndim; ++i) {
      p.info.shape[i] = shape[i];
      sz *= shape[i];
    }
    p.info.size = sz;
    p.info.offset = param_elems;
    p.info.trainable = trainable;
    p.info.backbone = backbone;
    param_elems += (sz + 63) & ~(int64_t)63;  // 256-B aligned tensors
    params.push_back(p);
    return (int)params.size() - 1;
  }

  int add_plane(int C, int div) {
    PlaneT p;
    p.C = C;
    p.div = div;
    planes.push_back(p);
    return (int)planes.size() - 1;
  }

  // `_conv2d` cell (layers.py:63-70) or a bare prediction conv (yolo3.py:62) when bn == false
  int add_conv(const std::string& name, int in_plane, int in_co, int cin, int out_plane, int out_co, int cout,
               int k, int stride, bool bn, int backbone, int res_plane = -1, int res_co = 0, int ups = 1) {
    ConvT c;
    c.name = name;
    c.in_plane = in_plane;
    c.in_co = in_co;
    c.cin = cin;
    c.out_plane = out_plane;
    c.out_co = out_co;
    c.cout = cout;
    c.k = k;
    c.stride = stride;
    c.leaky = bn ? 1 : 0;
    c.ups = ups;
    c.res_plane = res_plane;
    c.res_co = res_co;
    c.is_stem = (cin == 3);
    c.p_gamma = c.p_beta = c.p_mean = c.p_var = c.p_bias = -1;
    c.scale_off = c.shift_off = -1;
    const int wshape[4] = {cout, cin, k, k};
    const int cshape[1] = {cout};
    if (bn) {
      c.p_weight = add_param(name + ".0.weight", VY_P_WEIGHT, 4, wshape, 1, backbone);
      c.p_gamma = add_param(name + ".1.gamma", VY_P_GAMMA, 1, cshape, 1, backbone);
      c.p_beta = add_param(name + ".1.beta", VY_P_BETA, 1, cshape, 1, backbone);
      c.p_mean = add_param(name + ".1.running_mean", VY_P_RUNNING_MEAN, 1, cshape, 0, backbone);
      c.p_var = add_param(name + ".1.running_var", VY_P_RUNNING_VAR, 1, cshape, 0, backbone);
    } else {
      c.p_weight = add_param(name + ".weight", VY_P_WEIGHT, 4, wshape, 1, backbone);
      c.p_bias = add_param(name + ".bias", VY_P_BIAS, 1, cshape, 1, backbone);
    }
    convs.push_back(c);
    return (int)convs.size() - 1;
  }

  void build() {
    char nm[128];
    const int layers[5] = {1, 2, 8, 8, 4};
    const int chans[6] = {32, 64, 128, 256, 512, 1024};
    const int C = num_class;
    const int npred = 3 * (5 + C);
    // concat planes of the two shallower heads: [upsampled transition | backbone route]
    const int cat1 = add_plane(256 + 512, 16);  // yolo3.py:1177 at stride 16
    const int cat2 = add_plane(128 + 256, 8);   // ... at stride 8
    // ---- Darknet-53 features (three_darknet.py:162-195), named by stage slice (wrappers.py:58)
    int feat = 0;  // index into features[]
    auto feat_name = [&](int f) {
      int si = f < 15 ? 0 : (f < 24 ? 1 : 2);
      int j = f - (si == 0 ? 0 : (si == 1 ? 15 : 24));
      snprintf(nm, sizeof nm, "stages.%d.%d", si, j);
      return std::string(nm);
    };
    int cur = add_plane(32, 1), cur_co = 0;
    add_conv(feat_name(feat++), -1, 0, 3, cur, 0, 32, 3, 1, true, 1);
    int div = 1;
    for (int st = 0; st < 5; ++st) {
      const int ch = chans[st + 1];
      div *= 2;
      int nxt = add_plane(ch, div);
      add_conv(feat_name(feat++), cur, cur_co, chans[st], nxt, 0, ch, 3, 2, true, 1);
      cur = nxt;
      cur_co = 0;
      for (int bi = 0; bi < layers[st]; ++bi) {
        const std::string pre = feat_name(feat++);
        const int mid = add_plane(ch / 2, div);
        add_conv(pre + ".body.0", cur, cur_co, ch, mid, 0, ch / 2, 1, 1, true, 1);
        // the last block of stages 0 and 1 (features[14], features[23]) writes the route straight
        // into its concat plane
        int outp, outco = 0;
        if (feat - 1 == 14) {
          outp = cat2;
          outco = 128;
        } else if (feat - 1 == 23) {
          outp = cat1;
          outco = 256;
        } else {
          outp = add_plane(ch, div);
        }
        add_conv(pre + ".body.1", mid, 0, ch / 2, outp, outco, ch, 3, 1, true, 1, cur, cur_co);
        cur = outp;
        cur_co = outco;
      }
    }
    // ---- heads, deep -> shallow (yolo3.py:1013-1054, 1126-1177)
    const int hch[3] = {512, 256, 128};
    const int hdiv[3] = {32, 16, 8};
    int x = cur, x_co = cur_co, x_c = 1024;
    for (int i = 0; i < 3; ++i) {
      const int ch = hch[i], dv = hdiv[i];
      for (int j = 0; j < 5; ++j) {
        snprintf(nm, sizeof nm, "yolo_blocks.%d.body.%d", i, j);
        const int oc = (j % 2 == 0) ? ch : ch * 2;
        const int op = add_plane(oc, dv);
        add_conv(nm, x, x_co, x_c, op, 0, oc, (j % 2 == 0) ? 1 : 3, 1, true, 0);
        x = op;
        x_co = 0;
        x_c = oc;
      }
      const int route = x;
      snprintf(nm, sizeof nm, "yolo_blocks.%d.tip", i);
      const int tip = add_plane(ch * 2, dv);
      add_conv(nm, route, 0, ch, tip, 0, ch * 2, 3, 1, true, 0);
      snprintf(nm, sizeof nm, "yolo_outputs.%d.prediction", i);
      // channel stride padded to 32 so the plane can be the A operand of the prediction conv's dgrad
      head_plane[i] = add_plane((npred + 31) & ~31, dv);
      add_conv(nm, tip, 0, ch * 2, head_plane[i], 0, npred, 1, 1, false, 0);
      if (i == 2) break;
      // transition 1x1 (c -> c/2) stored x2-replicated into channels [0, c/2) of the concat plane
      snprintf(nm, sizeof nm, "transitions.%d", i);
      const int cat = (i == 0) ? cat1 : cat2;
      add_conv(nm, route, 0, ch, cat, 0, ch / 2, 1, 1, true, 0, -1, 0, 2);
      x = cat;
      x_co = 0;
      x_c = planes[cat].C;
    }
    tensor_elems = param_elems;
    // folded-BN scratch behind the tensors
    for (auto& c : convs) {
      if (c.p_gamma < 0) continue;
      c.scale_off = param_elems;
      param_elems += (c.cout + 63) & ~63;
      c.shift_off = param_elems;
      param_elems += (c.cout + 63) & ~63;
      FoldDesc f;
      f.gamma = params[c.p_gamma].info.offset;
      f.beta = params[c.p_beta].info.offset;
      f.mean = params[c.p_mean].info.offset;
      f.var = params[c.p_var].info.offset;
      f.scale = c.scale_off;
      f.shift = c.shift_off;
      f.C = c.cout;
      f.pad = 0;
      folds.push_back(f);
    }
  }

  // ---- planning
  static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

  // which inference launches the split-fp32 kernel takes in conv mode VY_CONV_SPLIT_BF16X3: every conv+BN+leaky cell
  // whose tile geometry it has (cout a multiple of 64, cin of 32) — the 3x3 cells (1.4-1.5x the exact kernel at
  // 608x608 batch 64; the two 64-channel ones 1.15x) and the 1x1 cells (1.13x at K = 128 ... 1.5x at K = 1024;
  // VY_SPLIT_1X1=0 keeps those exact).  The stem (Cin = 3), the 64 -> 32 bottleneck and the prediction convs
  // (75 channels) stay on the exact kernel.
  bool split_eligible(const ConvT& c) const {
    static const int with_1x1 = getenv("VY_SPLIT_1X1") ? atoi(getenv("VY_SPLIT_1X1")) : 1;
    if (conv_mode == VY_CONV_EXACT_FP32 || c.is_stem || c.p_gamma < 0) return false;
    if (c.cout % 64 != 0 || c.cin % 32 != 0) return false;
    return c.k == 3 || with_1x1;
  }

  // Plane -> storage slot.  keep_all (training, or vy_net_set_keep_activations): every plane has its own storage — the
  // backward pass (and the parity taps) read them all.  Otherwise planes are recycled by liveness: a plane may take over
  // the storage of an earlier plane of the SAME geometry (channel stride and resolution: the zero borders coincide and
  // stay zero, the interior is fully overwritten by the producing conv) once every reader of that plane has been
  // launched — strictly before the new plane's first writer, because a conv reads its input / addend while it writes.
  // In a residual stage that is two alternating block-output planes and one bottleneck plane instead of 2 x blocks.
  std::vector<int> plane_slots(bool keep_all) const {
    const int np = (int)planes.size();
    std::vector<int> slot(np);
    for (int i = 0; i < np; ++i) slot[i] = i;
    if (keep_all) return slot;
    const int kLive = 1 << 30;
    std::vector<int> def(np, kLive), last(np, -1);
    for (int ci = 0; ci < (int)convs.size(); ++ci) {
      const ConvT& c = convs[ci];
      if (c.out_plane >= 0 && ci < def[c.out_plane]) def[c.out_plane] = ci;
      if (c.in_plane >= 0 && ci > last[c.in_plane]) last[c.in_plane] = ci;
      if (c.res_plane >= 0 && ci > last[c.res_plane]) last[c.res_plane] = ci;
    }
    for (int i = 0; i < 3; ++i) last[head_plane[i]] = kLive;  // read by decode + NMS (and vy_net_read_head) afterwards
    std::vector<int> order(np);
    for (int i = 0; i < np; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return def[a] < def[b]; });
    std::vector<int> free_at(np, -1);  // per slot (= index of the plane that owns the storage): last reader so far
    std::vector<int> owners;
    for (int p : order) {
      int take = -1;
      // a prediction plane never takes over a freed plane: its conv writes npred of the C (padded) channels and the
      // padding must stay the zeros of the bind-time memset (80 classes: 255 -> 256 channels, the geometry of the
      // stage-1 bottleneck planes)
      const bool is_head = p == head_plane[0] || p == head_plane[1] || p == head_plane[2];
      for (int o : owners)
        if (!is_head && planes[o].C == planes[p].C && planes[o].div == planes[p].div && free_at[o] < def[p]) {
          take = o;
          break;
        }
      if (take < 0) {
        take = p;
        owners.push_back(p);
      }
      slot[p] = take;
      free_at[take] = last[p];
    }
    return slot;
  }

  size_t plan(int b, int h, int w, bool commit, bool keep_all) {
    size_t off = 0;
    const size_t fold_off = off;
    off += al(sizeof(FoldDesc) * folds.size());
    const size_t det_off = off;
    int n_items = 0;
    // a plane's spatial size = ceil(input / div): every stride-2 3x3 pad-1 conv maps n rows to ceil(n / 2), and the
    // x2 upsample is cropped to the route it is concatenated with (slice_like, yolo3.py:1177) — inputs need not be
    // multiples of 32
    auto cdiv = [](int a, int d) { return (a + d - 1) / d; };
    for (int i = 0; i < 3; ++i) n_items += 3 * cdiv(h, planes[head_plane[i]].div) * cdiv(w, planes[head_plane[i]].div);
    off += al(vy_det_scratch_bytes(b, n_items, num_class));
    const size_t sk_o = off;  // stream-K scratch of the conv launches (conv_igemm.hip): flags first, then the slabs
    off += al((size_t)VY_SK_FLAGS * sizeof(unsigned)) + al(VY_SK_PARTIAL_BYTES);
    // parked chains of the convs whose K is summed in runs (vy_conv_k_chunks; conv_igemm.hip `boundary`): runs - 1 tile-shaped
    // slabs per output tile of the conv that needs most
    const size_t ck_o = off;
    size_t ck_b = 0;
    for (const ConvT& c : convs) {
      const int runs = c.is_stem ? 1 : vy_conv_runs(c.k * c.k, (c.cin + 31) >> 5);
      if (runs < 2) continue;
      const PlaneT& op = planes[c.out_plane];
      const long long Mo = (long long)b * cdiv(h, op.div) * cdiv(w, op.div);
      ck_b = std::max(ck_b, vy_conv_chunk_scratch_bytes(Mo, c.cout, runs));
    }
    off += al(ck_b);
    const size_t wsp_off = off;  // split-fp32 weight images (conv mode VY_CONV_SPLIT_BF16X3 only)
    for (ConvT& c : convs) {
      const bool el = split_eligible(c);
      if (commit) c.split_off = el ? (int64_t)(off - wsp_off) : -1;
      if (el) off += al(vy_split_weight_bytes(c.cout, c.k * c.k, c.cin));
      const bool wel = el && c.k == 3 && c.stride == 1 && c.ups == 1 && c.cout % 128 == 0;
      if (commit) c.wino_off = wel ? (int64_t)(off - wsp_off) : -1;
      if (wel) off += al(vy_wino_weight_bytes(c.cout, c.cin));
    }
    const size_t pl_off = off;
    size_t fl = 0;
    const std::vector<int> slot = plane_slots(keep_all);
    std::vector<size_t> slot_off(planes.size(), 0);
    for (size_t i = 0; i < planes.size(); ++i) {
      PlaneT& p = planes[i];
      const int ph = cdiv(h, p.div), pw = cdiv(w, p.div);
      if ((size_t)slot[i] == i) {  // owns its storage (owners come first in definition order: see plane_slots)
        slot_off[i] = fl;
        fl += ((size_t)b * (ph + 2) * (pw + 2) * p.C + 63) & ~(size_t)63;
      }
    }
    for (size_t i = 0; i < planes.size() && commit; ++i) {
      PlaneT& p = planes[i];
      p.H = cdiv(h, p.div);
      p.W = cdiv(w, p.div);
      p.off = slot_off[slot[i]];
    }
    off += fl * sizeof(float);
    if (commit) {
      planes_shared = !keep_all;
      fold_desc_off = fold_off;
      det_scratch_off = det_off;
      sk_off = sk_o;
      ck_off = ck_o;
      ck_bytes = ck_b;
      wsplit_off = wsp_off;
      split_dirty = dsplit_dirty = wino_dirty = true;
      planes_off = pl_off;
      B = b;
      H = h;
      W = w;
    }
    return off;
  }

  float* plane_ptr(int i) const { return reinterpret_cast<float*>(dev_ws + planes_off) + planes[i].off; }

  int check_ready() const {
    if (!dev_params) return fail(VY_ERR_STATE, "parameters not bound (vy_net_bind_params)");
    if (!dev_ws) return fail(VY_ERR_STATE, "workspace not bound (vy_net_bind_workspace)");
    return 0;
  }

  // forward launch of conv `c`: reads plane views, writes `out_plane_ptr` (the activation plane, or
  // the raw-conv "z" plane in training)
  ConvArgs conv_args(const ConvT& c) const {
    ConvArgs a;
    memset(&a, 0, sizeof a);
    const PlaneT& ip = planes[c.in_plane];
    const PlaneT& op = planes[c.out_plane];
    a.in = plane_ptr(c.in_plane);
    a.w = dev_params + params[c.p_weight].info.offset;
    if (c.p_gamma >= 0) {
      a.scale = dev_params + c.scale_off;
      a.shift = dev_params + c.shift_off;
    } else {
      a.scale = nullptr;
      a.shift = dev_params + params[c.p_bias].info.offset;
    }
    a.res = c.res_plane >= 0 ? plane_ptr(c.res_plane) : nullptr;
    a.out = plane_ptr(c.out_plane);
    a.stats = nullptr;
    const int Ho = (ip.H + c.stride - 1) / c.stride, Wo = (ip.W + c.stride - 1) / c.stride;
    a.B = B;
    a.LH = Ho;
    a.LW = Wo;
    a.M = B * Ho * Wo;
    a.a_Hp = ip.H + 2;
    a.a_Wp = ip.W + 2;
    a.a_cs = ip.C;
    a.a_co = c.in_co;
    a.a_s = c.stride;
    a.a_oy = a.a_ox = 1;
    a.Kc = c.cin;
    a.ntaps = c.k * c.k;
    for (int t = 0; t < a.ntaps; ++t) {
      a.tap_dy[t] = (signed char)(c.k == 3 ? t / 3 - 1 : 0);
      a.tap_dx[t] = (signed char)(c.k == 3 ? t % 3 - 1 : 0);
      a.tap_w[t] = (unsigned char)t;
    }
    a.w_taps = c.k * c.k;
    a.w_cin = c.cin;
    a.w_cout = c.cout;
    a.N = c.cout;
    a.o_Hp = op.H + 2;  // ups == 2: the route's size, which may be one short of 2 x (Ho, Wo) — the store crops
    a.o_Wp = op.W + 2;
    a.o_cs = op.C;
    a.o_co = c.out_co;
    a.o_s = c.ups;
    a.o_oy = a.o_ox = 1;
    a.ups = c.ups;
    if (c.res_plane >= 0) {
      a.r_cs = planes[c.res_plane].C;
      a.r_co = c.res_co;
    }
    a.leaky = c.leaky;
    a.dgrad = 0;
    a.w_split = c.split_off >= 0 ? dev_ws + wsplit_off + c.split_off : nullptr;
    a.w_wino = c.wino_off >= 0 ? dev_ws + wsplit_off + c.wino_off : nullptr;
    a.env_split_always_p1 = env_split_always + 1;
    a.env_wino_mode_p1 = env_wino_mode + 1;
    a.cus = cus;
    a.ck_scratch = ck_bytes ? reinterpret_cast<float*>(dev_ws + ck_off) : nullptr;
    a.ck_bytes = ck_bytes;
    a.splitk_slabs = reinterpret_cast<float*>(dev_ws + sk_off + al((size_t)VY_SK_FLAGS * sizeof(unsigned)));
    a.splitk_bytes = VY_SK_PARTIAL_BYTES;
    set_sk(a);
    return a;
  }

  int sk_begin(hipStream_t s) {
    if (sk_dirty && dev_ws) {
      HIP_TRY(hipMemsetAsync(dev_ws + sk_off, 0, al((size_t)VY_SK_FLAGS * sizeof(unsigned)), s));
      sk_dirty = false;
    }
    return 0;
  }
  int sk_end(int rc) {
    if (rc != 0) sk_dirty = true;
    return rc;
  }

  // the stream-K scratch of this net's workspace (zeroed with the workspace at bind time: all flags down); left null —
  // plain launches only — unless the device's workgroup placement was verified
  void set_sk(ConvArgs& a) const {
    if (!sk_ok) return;
    a.sk_flags = reinterpret_cast<unsigned*>(dev_ws + sk_off);
    a.sk_partials = reinterpret_cast<float*>(dev_ws + sk_off + al((size_t)VY_SK_FLAGS * sizeof(unsigned)));
    a.sk_bytes = VY_SK_PARTIAL_BYTES;
    a.sk_nflags = VY_SK_FLAGS;
  }

  DetArgs det_args() const {
    DetArgs d;
    memset(&d, 0, sizeof d);
    int base = 0;
    for (int i = 0; i < 3; ++i) {
      const PlaneT& p = planes[head_plane[i]];
      HeadView& hv = d.head[i];
      hv.pred = plane_ptr(head_plane[i]);
      hv.H = p.H;
      hv.W = p.W;
      hv.cs = p.C;
      hv.co = 0;
      hv.stride = (float)kStrides[2 - i];  // anchors/strides used in reverse order, yolo3.py:1013
      for (int a = 0; a < 3; ++a) {
        hv.aw[a] = (float)kAnchors[2 - i][2 * a];
        hv.ah[a] = (float)kAnchors[2 - i][2 * a + 1];
      }
      hv.cand_base = base;
      base += num_class * p.H * p.W * 3;
    }
    d.B = B;
    d.C = num_class;
    d.n_cand = base;
    d.valid_thresh = 0.01f;  // yolo3.py:1199
    d.nms_thresh = nms_thresh;
    d.topk = nms_topk;
    d.post_nms = post_nms;
    d.do_nms = 1;
    return d;
  }

  // launches of one inference forward; `hook` (optional) is called around every launch
  // kLabels = false: the hook is a no-op (vy_net_forward_infer) and the per-launch labels are not built
  template <bool kLabels = true, typename Hook>
  int forward(const float* x, float* ids, float* scores, float* bboxes, int32_t* keep_idx, hipStream_t s,
              Hook&& hook) {
    if (int rc = check_ready()) return rc;
    refresh_env();
    const bool nms_on = nms_thresh > 0.f && nms_thresh < 1.f;  // yolo3.py:1197
    FoldDesc* fd = reinterpret_cast<FoldDesc*>(dev_ws + fold_desc_off);
    if (!fold_uploaded) {
      HIP_TRY(hipMemcpyAsync(fd, folds.data(), sizeof(FoldDesc) * folds.size(), hipMemcpyHostToDevice, s));
      HIP_TRY(hipStreamSynchronize(s));  // `folds` is pageable host memory; one-time
      fold_uploaded = true;
    }
    // The fold runs EVERY forward (the caller owns the parameter buffer and may have written to it) — inside the stem launch
    // when that has a block per layer (StemArgs), in a launch of its own otherwise.  VY_FOLD_IN_STEM=0: always its own.
    static const int fold_in_stem_on = getenv("VY_FOLD_IN_STEM") ? atoi(getenv("VY_FOLD_IN_STEM")) : 1;
    const bool fold_in_stem = fold_in_stem_on && !convs.empty() && convs[0].is_stem && convs[0].scale_off == folds[0].scale &&
                              vy_stem_can_fold(B, H, W, (int)folds.size());
    if (!fold_in_stem) {
      hook("bn_fold", 0.0, 0.0, true);
      HIP_TRY(vy_launch_bn_fold(dev_params, fd, (int)folds.size(), 1024, 1e-5f, s));
      hook("bn_fold", 0.0, 0.0, false);
    }
    if (conv_mode != VY_CONV_EXACT_FP32 && (split_dirty || wino_dirty)) {  // once per parameter change, not per forward
      hook("split_weights", 0.0, 0.0, true);
      for (const ConvT& c : convs)
        if (split_dirty && c.split_off >= 0)
          HIP_TRY(vy_launch_split_weights(dev_params + params[c.p_weight].info.offset, dev_ws + wsplit_off + c.split_off,
                                          c.cout, c.k * c.k, c.cin, s));
      for (const ConvT& c : convs)
        if (wino_dirty && c.wino_off >= 0)
          HIP_TRY(vy_launch_wino_weights(dev_params + params[c.p_weight].info.offset, dev_ws + wsplit_off + c.wino_off, c.cout,
                                         c.cin, s));
      hook("split_weights", 0.0, 0.0, false);
      split_dirty = wino_dirty = false;
    }
    for (const ConvT& c : convs) {
      if (c.is_stem) {
        StemArgs a;
        a.x = x;
        a.w = dev_params + params[c.p_weight].info.offset;
        a.scale = dev_params + c.scale_off;
        a.shift = dev_params + c.shift_off;
        a.out = plane_ptr(c.out_plane);
        a.B = B;
        a.H = H;
        a.W = W;
        a.Cout = c.cout;
        a.out_cs = planes[c.out_plane].C;
        a.out_co = c.out_co;
        if (fold_in_stem) {
          a.fold_params = dev_params;
          a.fold_descs = fd;
          a.fold_n = (int)folds.size();
          a.fold_stem = 0;
          a.fold_eps = 1e-5f;
        }
        const double fl = 2.0 * B * H * W * 27.0 * c.cout;
        const double by = 4.0 * B * H * W * (3.0 + c.cout);
        hook(c.name.c_str(), fl, by, true);
        HIP_TRY(vy_launch_stem(a, s));
        hook(c.name.c_str(), fl, by, false);
      } else {
        const ConvArgs a = conv_args(c);
        if (!kLabels) {  // plain forward: no label, no second tile / stream-K query per launch (batch-1 latency path)
          if (a.w_wino && vy_conv_wino_pays(a))
            HIP_TRY(vy_launch_conv_wino(a, s));
          else if (a.w_split && vy_conv_split_pays(a))
            HIP_TRY(vy_launch_conv_split(a, s));
          else
            HIP_TRY(vy_launch_conv_igemm(a, s));
          continue;
        }
        const double fl = 2.0 * a.M * (double)a.N * a.ntaps * a.Kc;
        const double by = 4.0 * ((double)B * (a.a_Hp - 2) * (a.a_Wp - 2) * a.Kc + (double)a.M * a.N * a.ups * a.ups +
                                 (double)a.N * a.ntaps * a.Kc + (a.res ? (double)a.M * a.N : 0.0));
        char nm[96];
        if (a.w_wino && vy_conv_wino_pays(a)) {
          snprintf(nm, sizeof nm, "%s|wino64x128", c.name.c_str());
          hook(nm, fl, by, true);
          HIP_TRY(vy_launch_conv_wino(a, s));
          hook(nm, fl, by, false);
          continue;
        }
        if (a.w_split && vy_conv_split_pays(a)) {
          int sbm, sbn, sks;
          vy_conv_split_cfg(a, &sbm, &sbn, &sks);
          if (sks > 1) snprintf(nm, sizeof nm, "%s|split%dx%dk%d", c.name.c_str(), sbm, sbn, sks);
          else snprintf(nm, sizeof nm, "%s|split%dx%d", c.name.c_str(), sbm, sbn);
          hook(nm, fl, by, true);
          HIP_TRY(vy_launch_conv_split(a, s));
          hook(nm, fl, by, false);
          continue;
        }
        int bm, bn;
        vy_conv_cfg(a, &bm, &bn);
        if (const int ks = vy_conv_ksplit(a))
          snprintf(nm, sizeof nm, "%s|%dx%dks%d", c.name.c_str(), bm, bn, ks);
        else
          snprintf(nm, sizeof nm, "%s|%dx%d%s", c.name.c_str(), bm, bn, vy_conv_streamk(a) ? "sk" : "");
        hook(nm, fl, by, true);
        HIP_TRY(vy_launch_conv_igemm(a, s));
        hook(nm, fl, by, false);
      }
    }
    const DetArgs d = det_args();
    double dby = 0;
    for (int i = 0; i < 3; ++i) dby += 4.0 * B * d.head[i].H * d.head[i].W * d.head[i].cs;
    hook("decode_nms", 0.0, dby, true);
    if (nms_on)
      HIP_TRY(vy_launch_detect(d, dev_ws + det_scratch_off, ids, scores, bboxes, keep_idx, s));
    else  // outputs are (B, N*C, .): the detection tensor itself
      HIP_TRY(vy_launch_raw_detections(d, ids, scores, bboxes, keep_idx, s));
    hook("decode_nms", 0.0, dby, false);
    return 0;
  }
};

