// train.hip — host side of the training step: plans the extra planes (raw conv outputs z / their
// gradients dz, gradient planes g), builds the forward-train and backward launch sequences from the
// same conv list the inference path uses, and implements the training C-ABI of include/vyolo.h.
//
// Forward (recording), per `_conv2d` cell (layers.py:63-70 under autograd.record()):
//   conv (raw, + per-tile channel sums) -> reduce partials -> [SyncBN all-reduce] -> finalize
//   (mean/var -> scale/shift, running stats) -> apply (BN affine + leaky [+ residual] [x2 replicate])
// then the prediction convs, then the fused target-merge + loss + d(loss)/d(pred) kernel.
// Backward, cells in reverse order:
//   BN/leaky backward reduce -> [all-reduce] -> finalize (dgamma, dbeta, coefficients) -> apply (dz
//   overwrites z) -> wgrad (split-K slabs + ordered reduce) -> dgrad into the input's gradient plane
//   (overwrite, or accumulate when the plane already holds another consumer's contribution, plus
//   the skip connection's gradient as an addend).
#include <cstdlib>

#include "net_internal.h"

namespace {

struct ZPlane {
  int C = 0;        // channel stride (Cout padded to 4; 0 = none)
  size_t off = 0;   // float offset in the z region
  int H = 0, W = 0;
};

struct BwdDgrad {
  ConvArgs a[4];
  int n = 0;
};

}  // namespace

struct VyTrain {
  int B = 0, H = 0, W = 0;
  float* grads = nullptr;
  float* mom = nullptr;
  float ignore_iou = 0.7f;
  int label_smooth = 0;
  // regions (byte offsets in the workspace)
  size_t g_off = 0, z_off = 0, save_off = 0, coef_off = 0, sums_off = 0, slice_off = 0, part_off = 0, slab_off = 0;
  size_t loss_part_off = 0, loss_off = 0, zero_off = 0, seg_off = 0, chunk_off = 0, total = 0;
  size_t part_floats = 0, slab_floats = 0;
  std::vector<ZPlane> z;               // per conv
  std::vector<size_t> save_idx;        // per conv: float offset of [2][C] saved mean/invstd
  std::vector<int> splits, kps;        // per conv wgrad split-K
  // conv mode VY_CONV_SPLIT_BF16X3: per conv, byte offset of its weights' DATA-GRADIENT tile images ([k = cout][n = cin]
  // operand; conv_split.hip) inside the region at dsplit_off, -1: exact kernel.  The forward images are the net's.
  std::vector<long long> dsplit;
  size_t dsplit_off = 0;
  std::vector<SplitDesc> sdesc;        // all image sets (forward + data gradient) for the one-launch rebuild
  size_t sdesc_off = 0;
  long long sdesc_total = 0;
  bool sdesc_uploaded = false;
  std::vector<size_t> tab_off;         // per conv: byte offset of its weight-gradient pixel table in the workspace
  bool tabs_built = false;
  std::vector<SgdSeg> segs;
  std::vector<int32_t> chunk_seg;
  bool seg_uploaded = false;
  std::vector<float> lr_mult, wd_mult;
  std::vector<int> enabled;
  // SyncBN
  int world = 1;
  vy_allreduce_cb ar_cb = nullptr;
  void* ar_user = nullptr;
  vy_grad_bucket_cb gb_cb = nullptr;
  void* gb_user = nullptr;
  bool forward_done = false;
  int M = 0;
  // weight gradients run on a side stream, concurrently with the dgrad / BatchNorm chain
  hipStream_t side = nullptr;
  hipEvent_t ev_main = nullptr, ev_side = nullptr;
  ~VyTrain() {
    if (ev_main) (void)hipEventDestroy(ev_main);
    if (ev_side) (void)hipEventDestroy(ev_side);
    if (side) (void)hipStreamDestroy(side);
  }
};

void vy_train_free(vy_net* net) {
  delete net->train;
  net->train = nullptr;
}

namespace {

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// Profiling aid (tools/train_layers.py): with VY_TRAIN_LABELS=<path> the first training step appends one line per
// matrix-core launch — kind (fwd / wgrad / dgrad), cell name, FLOPs, GEMM dims — so that a rocprofv3 kernel trace
// can be joined with the layers by launch order within each kernel class.
struct LabelLog {
  FILE* f = nullptr;
  bool open_once() {
    static const char* path = getenv("VY_TRAIN_LABELS");
    if (!path) return false;
    if (!f) f = fopen(path, "w");
    return f != nullptr;
  }
  void note(const char* kind, const std::string& name, double M, double N, double K) {
    if (open_once()) fprintf(f, "%s %s %.0f %.0f %.0f %.0f\n", kind, name.c_str(), 2.0 * M * N * K, M, N, K);
  }
  // "# via split 128x128 k2" / "# via exact": which kernel the preceding fwd / dgrad line went to (tests assert that the
  // launches they mean to cover really ran; tools/train_layers.py skips '#' lines)
  void via(const char* text) {
    if (f) fprintf(f, "# via %s\n", text);
  }
  void close_step() {
    if (f) {
      fclose(f);
      f = nullptr;
    }
  }
};
LabelLog g_labels;
bool g_labels_done = false;

// Measurement aid (tools/ab_bn_bounds.sh): VY_TRAIN_ABL skips BatchNorm launches of the training step to BOUND what
// fusing them into the neighbouring conv launches could return — bit 1: bn_bwd_reduce (+ its finalize), 2: the forward
// bn_apply, 4: bn_bwd_apply; and, to see which stream of the backward pass holds the step, 8: no weight-gradient
// kernels, 16: no data-gradient kernels, 32: no weight-gradient kernels for the early cells (N <= 128, K <= 576); 64 / 128: the
// finalize launch between the backward / forward statistics reduce and its apply pass (tools/ab_bn_finalize.sh).  The step then
// computes garbage; nothing else reads this.
// Compiled in ONLY with -DVY_TRAIN_ABL_BUILD (the A/B scripts build their own library): the shipped library never reads
// the variable, so a leftover VY_TRAIN_ABL in somebody's environment cannot silently turn a training run into garbage.
#ifdef VY_TRAIN_ABL_BUILD
// bits 64 / 128 (a finalize launch skipped: its outputs keep the previous step's values) only after VY_TRAIN_ABL_AFTER recorded
// forwards (default 3), so that the planes never hold the all-zero data of a net whose statistics were never finalized
static int g_abl_forwards = 0;
static int train_abl_all();
static int train_abl() {
  static const int after = getenv("VY_TRAIN_ABL_AFTER") ? atoi(getenv("VY_TRAIN_ABL_AFTER")) : 3;
  const int v = train_abl_all();
  return g_abl_forwards > after ? v : (v & ~(64 | 128));
}
static int train_abl_all() {
  static const int v = [] {
    const int e = getenv("VY_TRAIN_ABL") ? atoi(getenv("VY_TRAIN_ABL")) : 0;
    if (e) fprintf(stderr, "libvyolo (VY_TRAIN_ABL_BUILD): VY_TRAIN_ABL=%d — training launches are being SKIPPED, gradients are garbage\n", e);
    return e;
  }();
  return v;
}
#else
static constexpr int train_abl() { return 0; }
#endif

constexpr int kBwdChunk = 64;  // pixels per partial-sum block of the bias-gradient reductions (and the scratch bound)

VyTrain* get_train(vy_net* net) {
  if (!net->train) {
    net->train = new VyTrain();
    VyTrain* t = net->train;
    t->lr_mult.assign(net->params.size(), 1.0f);
    t->wd_mult.assign(net->params.size(), 1.0f);
    t->enabled.resize(net->params.size());
    for (size_t i = 0; i < net->params.size(); ++i) t->enabled[i] = net->params[i].info.trainable;
  }
  return net->train;
}

// plan the training regions behind the inference workspace
size_t train_plan(vy_net* net, int b, int h, int w, bool commit) {
  VyTrain* t = get_train(net);
  size_t off = al256(net->plan(b, h, w, commit, /*keep_all=*/true));  // backward reads every activation plane
  // split-fp32 conv mode: the data gradients whose N (= cin) the split kernel has a tile for get their own weight images
  std::vector<long long> dsplit(net->convs.size(), -1);
  const size_t dsplit_off = off;
  if (net->conv_mode == VY_CONV_SPLIT_BF16X3_TRAIN) {
    static const int train_split = getenv("VY_SPLIT_TRAIN") ? atoi(getenv("VY_SPLIT_TRAIN")) : 1;
    for (size_t i = 0; i < net->convs.size() && train_split; ++i) {
      const ConvT& c = net->convs[i];
      if (c.is_stem || c.cin % 64 != 0) continue;
      dsplit[i] = (long long)(off - dsplit_off);
      off += al256(vy_split_weight_dgrad_bytes(c.cout, c.k * c.k, c.cin));
    }
  }
  // gradient planes mirror the activation planes
  size_t gfl = 0;
  for (auto& p : net->planes) gfl += ((size_t)b * (h / p.div + 2) * (w / p.div + 2) * p.C + 63) & ~(size_t)63;
  const size_t g_off = off;
  off += al256(gfl * sizeof(float));
  // z planes: one per BN conv
  std::vector<ZPlane> z(net->convs.size());
  std::vector<size_t> save(net->convs.size(), 0);
  std::vector<int> splits(net->convs.size(), 1), kps(net->convs.size(), 32);
  size_t zfl = 0, sfl = 0, part = 0, slab = 0;
  for (size_t i = 0; i < net->convs.size(); ++i) {
    const ConvT& c = net->convs[i];
    const int div_in = c.is_stem ? 1 : net->planes[c.in_plane].div;
    const int Ho = h / div_in / c.stride, Wo = w / div_in / c.stride;
    const long long M = (long long)b * Ho * Wo;
    if (c.p_gamma >= 0) {
      z[i].C = c.cout;
      z[i].H = Ho;
      z[i].W = Wo;
      z[i].off = zfl;
      zfl += ((size_t)b * (Ho + 2) * (Wo + 2) * c.cout + 63) & ~(size_t)63;
      save[i] = sfl;
      sfl += 2 * (size_t)((c.cout + 63) & ~63);
      // partials: forward stats, backward sums
      const size_t tiles_m = (size_t)((M + 31) / 32);  // per-tile statistics rows: the smallest tile height (conv_small.hip)
      size_t pf = 2 * (c.is_stem ? (size_t)vy_stem_blocks(b, h, w) * 64 : tiles_m * 2 * c.cout);  // doubles
      const size_t chunks = (size_t)((M + kBwdChunk - 1) / kBwdChunk);
      if (chunks * 2 * c.cout > pf) pf = chunks * 2 * c.cout;
      // bn_bwd_reduce chunks by image rows, not by 64 pixels: ceil(B*Ho / rows_per_chunk) partial rows of 2*C
      // floats — more than the pixel-chunk bound on maps narrower than 64 pixels (W = 32 training shapes)
      {
        const int rpc = vy_bn_bwd_rows_per_chunk(b, Ho, c.cout);
        const size_t bwd_rows = (size_t)(((long long)b * Ho + rpc - 1) / rpc);
        if (bwd_rows * 2 * c.cout > pf) pf = bwd_rows * 2 * c.cout;
      }
      if (c.is_stem) {
        const size_t sw = (size_t)vy_stem_wgrad_blocks(b, h, w) * 864;
        if (sw > pf) pf = sw;
      }
      if (pf > part) part = pf;
    } else {
      const size_t chunks = (size_t)((M + kBwdChunk - 1) / kBwdChunk);
      if (chunks * c.cout > part) part = chunks * c.cout;
    }
    if (!c.is_stem) {
      const int Ntot = c.k * c.k * c.cin;
      const int rows = vy_wgrad_tile_rows(c.cout, c.k, c.cin);
      const int tiles = ((c.cout + rows - 1) / rows) * ((Ntot + 127) / 128);
      // Split-K over the pixels.  A block runs k_per_split / 32 k-steps (+ ~3 k-steps worth of prologue and slab
      // store); the chip holds 512 blocks at a time (2 per CU), so the launch costs about
      // ceil(tiles * splits / 512) * (k_per_split / 32 + 3).  Round 1 took the smallest split count with >= 1024
      // blocks, which for the three big 3x3 groups lands just past a multiple of 512 (18 x 57 = 1026,
      // 72 x 15 = 1080, 288 x 4 = 1152 blocks): a last round of a few blocks with the chip idle around them.  Now the
      // cheapest (splits, k_per_split) under that model is taken; a split is >= 256 pixels.
      long long best_k = ((M + 31) / 32) * 32, best_sp = 1;
      double best_cost = 1e300;
      const long long held = 2ll * net->resolve_cus();  // blocks the net's device holds at a time (512 on the MI355X)
      for (long long k = 256; k <= ((M + 31) / 32) * 32; k += 32) {
        const long long sp_k = (M + k - 1) / k;
        const long long rounds = (tiles * sp_k + held - 1) / held;
        const double cost = (double)rounds * ((double)k / 32.0 + 3.0);
        if (cost < best_cost) {
          best_cost = cost;
          best_k = k;
          best_sp = sp_k;
        }
      }
      if (M < 256) {
        best_k = ((M + 31) / 32) * 32;
        best_sp = 1;
      }
      const long long sp = best_sp;
      splits[i] = (int)sp;
      kps[i] = (int)best_k;
      const size_t sl = (size_t)sp * c.cout * Ntot;
      if (sl > slab) slab = sl;
    }
  }
  const size_t z_off = off;
  off += al256(zfl * sizeof(float));
  const size_t save_off = off;
  off += al256(sfl * sizeof(float));
  const size_t coef_off = off;
  off += al256(3 * 1024 * sizeof(float));
  const size_t sums_off = off;
  off += al256(2 * 2 * 1024 * sizeof(double));  // [global | local] x [2][C]
  const size_t slice_off = off;
  off += al256((size_t)VY_REDUCE_SLICES * 2 * 1024 * sizeof(double));  // slice sums of long per-tile statistics lists
  const size_t part_off = off;
  off += al256(part * sizeof(float));
  const size_t slab_off = off;
  off += al256(slab * sizeof(float));
  int N = 0;
  for (int i = 0; i < 3; ++i) {
    const int dv = net->planes[net->head_plane[i]].div;
    N += 3 * (h / dv) * (w / dv);
  }
  const size_t loss_part_off = off;
  off += al256((size_t)vy_loss_blocks_per_image(N) * b * 4 * sizeof(float));
  const size_t loss_off = off;
  off += al256((size_t)4 * b * sizeof(float));
  const size_t zero_off = off;
  off += 1024;
  // SGD segment tables
  std::vector<SgdSeg> segs;
  std::vector<int32_t> chunk_seg;
  for (size_t i = 0; i < net->params.size(); ++i) {
    const vy_param_info& pi = net->params[i].info;
    SgdSeg sg;
    sg.off = pi.offset;
    sg.size = pi.size;
    sg.lr_mult = t->lr_mult[i];
    sg.wd_mult = t->wd_mult[i];
    sg.enabled = (pi.trainable && t->enabled[i]) ? 1 : 0;
    sg.pad = 0;
    if (!pi.trainable) continue;
    const int32_t si = (int32_t)segs.size();
    segs.push_back(sg);
    const int nch = (int)((pi.size + VY_SGD_CHUNK - 1) / VY_SGD_CHUNK);
    for (int c = 0; c < nch; ++c) {
      chunk_seg.push_back(si);
      chunk_seg.push_back(c);
    }
  }
  // weight-gradient pixel tables (wgrad.hip): 8 bytes per output pixel of every conv but the stem
  std::vector<size_t> tab_off(net->convs.size(), 0);
  for (size_t i = 0; i < net->convs.size(); ++i) {
    const ConvT& c = net->convs[i];
    if (c.is_stem) continue;
    const int div_in = net->planes[c.in_plane].div;
    const long long M = (long long)b * (h / div_in / c.stride) * (w / div_in / c.stride);
    tab_off[i] = off;
    off += al256(vy_wgrad_table_entries(M) * 8);
  }
  const size_t sdesc_off = off;
  off += al256(sizeof(SplitDesc) * 2 * net->convs.size());
  const size_t seg_off = off;
  off += al256(segs.size() * sizeof(SgdSeg));
  const size_t chunk_off = off;
  off += al256(chunk_seg.size() * sizeof(int32_t));
  if (commit) {
    t->B = b;
    t->H = h;
    t->W = w;
    t->g_off = g_off;
    t->z_off = z_off;
    t->save_off = save_off;
    t->coef_off = coef_off;
    t->sums_off = sums_off;
    t->slice_off = slice_off;
    t->part_off = part_off;
    t->slab_off = slab_off;
    t->loss_part_off = loss_part_off;
    t->loss_off = loss_off;
    t->zero_off = zero_off;
    t->seg_off = seg_off;
    t->chunk_off = chunk_off;
    t->part_floats = part;
    t->slab_floats = slab;
    t->z = z;
    t->save_idx = save;
    t->splits = splits;
    t->kps = kps;
    t->dsplit = dsplit;
    t->dsplit_off = dsplit_off;
    t->sdesc.clear();
    t->sdesc_total = 0;
    for (size_t i = 0; i < net->convs.size(); ++i) {
      const ConvT& c = net->convs[i];
      const long long w_off = net->params[c.p_weight].info.offset;
      if (c.split_off >= 0) {
        t->sdesc.push_back(SplitDesc{t->sdesc_total, w_off, (long long)(net->wsplit_off + c.split_off), c.cout, c.k * c.k, c.cin, 0});
        t->sdesc_total += (long long)c.cout * c.k * c.k * c.cin / 8;
      }
      if (dsplit[i] >= 0) {
        t->sdesc.push_back(SplitDesc{t->sdesc_total, w_off, (long long)(dsplit_off + dsplit[i]), c.cout, c.k * c.k, c.cin, 1});
        t->sdesc_total += (long long)((c.cout + 31) & ~31) * c.k * c.k * c.cin / 8;
      }
    }
    t->sdesc_off = sdesc_off;
    t->sdesc_uploaded = false;
    t->tab_off = tab_off;
    t->tabs_built = false;
    t->segs = segs;
    t->chunk_seg = chunk_seg;
    t->seg_uploaded = false;
    t->total = off;
  }
  return off;
}

struct TrainCtx {
  vy_net* net;
  VyTrain* t;
  hipStream_t s;
  float* gplane(int i) const { return reinterpret_cast<float*>(net->dev_ws + t->g_off) + net->planes[i].off; }
  float* zplane(int ci) const { return reinterpret_cast<float*>(net->dev_ws + t->z_off) + t->z[ci].off; }
  float* save(int ci) const { return reinterpret_cast<float*>(net->dev_ws + t->save_off) + t->save_idx[ci]; }
  float* coef() const { return reinterpret_cast<float*>(net->dev_ws + t->coef_off); }
  double* sums_global() const { return reinterpret_cast<double*>(net->dev_ws + t->sums_off); }
  double* sums_local() const { return sums_global() + 2 * 1024; }
  double* slice_sums() const { return reinterpret_cast<double*>(net->dev_ws + t->slice_off); }
  float* partials() const { return reinterpret_cast<float*>(net->dev_ws + t->part_off); }
  float* slabs() const { return reinterpret_cast<float*>(net->dev_ws + t->slab_off); }
  const float* zero() const { return reinterpret_cast<const float*>(net->dev_ws + t->zero_off); }
  float* grad_of(int pidx) const { return t->grads + net->params[pidx].info.offset; }
};

// SyncBatchNorm statistics are exchanged when there is more than one rank — or when a callback was installed for ONE rank
// (videoyolo_amd.parallel with VY_FORCE_COLLECTIVES=1: the all-reduce over one rank is the identity; it exists so that the
// whole exchange path, RCCL included, can be executed on a one-GPU box)
static bool sync_exchange(const VyTrain* t) { return t->world > 1 || (t->world == 1 && t->ar_cb != nullptr); }

// the sums the normalisation uses: the local ones, or (SyncBN layers, world > 1) their all-reduce
int combine_sums(const TrainCtx& c, const ConvT& cv, int n_cols, double* count, const double** use) {
  *use = c.sums_local();
  if (sync_exchange(c.t) && is_sync_layer(cv)) {
    if (!c.t->ar_cb) return fail(VY_ERR_STATE, "SyncBN world > 1 without an all-reduce callback");
    HIP_TRY(hipMemcpyAsync(c.sums_global(), c.sums_local(), sizeof(double) * n_cols, hipMemcpyDeviceToDevice, c.s));
    if (int rc = c.t->ar_cb(c.t->ar_user, c.sums_global(), n_cols))
      return fail(VY_ERR_STATE, "all-reduce callback failed (%d)", rc);
    *count *= c.t->world;
    *use = c.sums_global();
  }
  return 0;
}

// conv mode VY_CONV_SPLIT_BF16X3 in training: both sets of weight images follow the parameters (rebuilt after every
// optimizer step: 0.6 GB of traffic, ~0.3 ms, beside a 26 ms step)
int refresh_split_images(const TrainCtx& c) {
  vy_net* net = c.net;
  if (net->conv_mode != VY_CONV_SPLIT_BF16X3_TRAIN || !(net->split_dirty || net->dsplit_dirty) || c.t->sdesc.empty()) return 0;
  SplitDesc* d = reinterpret_cast<SplitDesc*>(net->dev_ws + c.t->sdesc_off);
  if (!c.t->sdesc_uploaded) {
    HIP_TRY(hipMemcpyAsync(d, c.t->sdesc.data(), sizeof(SplitDesc) * c.t->sdesc.size(), hipMemcpyHostToDevice, c.s));
    HIP_TRY(hipStreamSynchronize(c.s));  // pageable host vector; once per plan
    c.t->sdesc_uploaded = true;
  }
  // one launch for every image set of the net (per-conv launches: 140 kernel boundaries per step)
  HIP_TRY(vy_launch_split_weights_batch(net->dev_params, net->dev_ws, d, (int)c.t->sdesc.size(), c.t->sdesc_total, c.s));
  net->split_dirty = net->dsplit_dirty = false;
  return 0;
}

// the conv launch of the training passes: the split-fp32 kernel where the net's mode allows it, the launch has its
// weight images (a.w_split) and the cost model predicts a gain; the exact kernel otherwise
static int launch_conv(const ConvArgs& a, hipStream_t s) {
  if (a.w_split && vy_conv_split_pays(a)) {
    if (!g_labels_done && g_labels.f) {
      int bm, bn, ks;
      char t[64];
      vy_conv_split_cfg(a, &bm, &bn, &ks);
      snprintf(t, sizeof t, "split %dx%d k%d", bm, bn, ks);
      g_labels.via(t);
    }
    HIP_TRY(vy_launch_conv_split(a, s));
  } else {
    if (!g_labels_done && g_labels.f) g_labels.via("exact");
    HIP_TRY(vy_launch_conv_igemm(a, s));
  }
  return 0;
}

int forward_train(const TrainCtx& c, const float* x) {
  vy_net* net = c.net;
  const int B = net->B;
  net->refresh_env();
  if (int rc = refresh_split_images(c)) return rc;
  for (size_t ci = 0; ci < net->convs.size(); ++ci) {
    const ConvT& cv = net->convs[ci];
    if (cv.p_gamma < 0) {  // prediction conv: bias, no BN
      const ConvArgs a = net->conv_args(cv);
      if (!g_labels_done) g_labels.note("fwd", cv.name, a.M, a.N, (double)a.ntaps * a.Kc);
      HIP_TRY(vy_launch_conv_igemm(a, c.s));
      continue;
    }
    const ZPlane& zp = c.t->z[ci];
    int n_part;
    if (cv.is_stem) {
      StemArgs a;
      a.x = x;
      a.w = net->dev_params + net->params[cv.p_weight].info.offset;
      a.scale = a.shift = nullptr;
      a.out = c.zplane((int)ci);
      a.B = B;
      a.H = net->H;
      a.W = net->W;
      a.Cout = cv.cout;
      a.out_cs = zp.C;
      a.out_co = 0;
      HIP_TRY(vy_launch_stem_raw(a, reinterpret_cast<double*>(c.partials()), c.s));
      n_part = vy_stem_blocks(B, net->H, net->W);
    } else {
      ConvArgs a = net->conv_args(cv);
      a.scale = a.shift = a.res = nullptr;
      a.leaky = 0;
      a.out = c.zplane((int)ci);
      a.o_Hp = zp.H + 2;
      a.o_Wp = zp.W + 2;
      a.o_cs = zp.C;
      a.o_co = 0;
      a.o_s = 1;
      a.ups = 1;
      a.stats = reinterpret_cast<double*>(c.partials());
      static const int train_split = getenv("VY_SPLIT_TRAIN") ? atoi(getenv("VY_SPLIT_TRAIN")) : 1;  // 0 none, 1 both, 2 forward only, 3 dgrad only
      if (net->conv_mode != VY_CONV_SPLIT_BF16X3_TRAIN || train_split == 0 || train_split == 3) a.w_split = nullptr;
      if (!g_labels_done) g_labels.note("fwd", cv.name, a.M, a.N, (double)a.ntaps * a.Kc);
      if (int rc = launch_conv(a, c.s)) return rc;
      if (a.w_split && vy_conv_split_pays(a)) {  // the per-tile statistics rows follow the tile that ran
        int sbm, sbn, sks;
        vy_conv_split_cfg(a, &sbm, &sbn, &sks);
        n_part = (a.M + sbm - 1) / sbm;
      } else {
        n_part = vy_conv_tiles_m(a);
      }
    }
    const int C = cv.cout;
    // statistics exchange between ranks only for the SyncBatchNorm layers; everywhere else the ordered
    // reduce of the per-tile sums and the finalize are one launch
    const bool exchange = sync_exchange(c.t) && is_sync_layer(cv);
    double count = (double)B * zp.H * zp.W;
    const double* use_sums = nullptr;
    if (exchange) {
      HIP_TRY(vy_launch_reduce_partials_f64(reinterpret_cast<const double*>(c.partials()), n_part, 2 * C,
                                            c.sums_local(), c.s));
      if (int rc = combine_sums(c, cv, 2 * C, &count, &use_sums)) return rc;
    }
    BnFinalizeArgs f;
    f.sums = use_sums;
    f.count = count;
    f.gamma = net->dev_params + net->params[cv.p_gamma].info.offset;
    f.beta = net->dev_params + net->params[cv.p_beta].info.offset;
    f.running_mean = net->dev_params + net->params[cv.p_mean].info.offset;
    f.running_var = net->dev_params + net->params[cv.p_var].info.offset;
    f.scale = net->dev_params + cv.scale_off;
    f.shift = net->dev_params + cv.shift_off;
    f.save_mean = c.save((int)ci);
    f.save_invstd = c.save((int)ci) + ((C + 63) & ~63);
    f.C = C;
    f.eps = 1e-5f;
    f.momentum = 0.9f;  // layers.py:68
    if (exchange)
      HIP_TRY(vy_launch_bn_finalize(f, c.s));
    else if (!(train_abl() & 128))  // (128: the same for the forward statistics)
      HIP_TRY(vy_launch_bn_reduce_finalize(reinterpret_cast<const double*>(c.partials()), n_part, f, c.slice_sums(), c.s));
    BnApplyArgs ap;
    memset(&ap, 0, sizeof ap);
    ap.z = c.zplane((int)ci);
    ap.scale = f.scale;
    ap.shift = f.shift;
    const PlaneT& op = net->planes[cv.out_plane];
    ap.out = net->plane_ptr(cv.out_plane);
    ap.B = B;
    ap.H = zp.H;
    ap.W = zp.W;
    ap.C = C;
    ap.o_Hp = zp.H * cv.ups + 2;
    ap.o_Wp = zp.W * cv.ups + 2;
    ap.o_cs = op.C;
    ap.o_co = cv.out_co;
    ap.ups = cv.ups;
    if (cv.res_plane >= 0) {
      ap.res = net->plane_ptr(cv.res_plane);
      ap.r_cs = net->planes[cv.res_plane].C;
      ap.r_co = cv.res_co;
    }
    if (!(train_abl() & 2)) HIP_TRY(vy_launch_bn_apply(ap, c.s));
  }
  return 0;
}

// dgrad launches of conv cv: gradient w.r.t. its input view, from dz stored in `dzp`
BwdDgrad make_dgrad(const TrainCtx& c, const ConvT& cv, const float* dzp, int dz_cs, int dz_H, int dz_W,
                    const float* addend, int add_cs, int add_co) {
  vy_net* net = c.net;
  BwdDgrad out;
  const PlaneT& ip = net->planes[cv.in_plane];
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.in = dzp;
  a.w = net->dev_params + net->params[cv.p_weight].info.offset;
  a.out = c.gplane(cv.in_plane);
  a.res = addend;
  a.r_cs = add_cs;
  a.r_co = add_co;
  a.B = net->B;
  a.a_Hp = dz_H + 2;
  a.a_Wp = dz_W + 2;
  a.a_cs = dz_cs;
  a.a_co = 0;
  a.a_s = 1;
  a.a_oy = a.a_ox = 1;
  a.Kc = (cv.cout + 31) & ~31;
  a.w_taps = cv.k * cv.k;
  a.w_cin = cv.cin;
  a.w_cout = cv.cout;
  a.N = cv.cin;
  a.o_Hp = ip.H + 2;
  a.o_Wp = ip.W + 2;
  a.o_cs = ip.C;
  a.o_co = cv.in_co;
  a.ups = 1;
  a.dgrad = 1;
  a.env_split_always_p1 = net->env_split_always + 1;  // (as read by this step's forward)
  a.env_wino_mode_p1 = net->env_wino_mode + 1;
  a.cus = net->cus;
  net->set_sk(a);
  {  // split-fp32 conv mode: this conv's data-gradient weight images and the split-K scratch (the stream-K region)
    const size_t ci = (size_t)(&cv - net->convs.data());
    static const int train_split = getenv("VY_SPLIT_TRAIN") ? atoi(getenv("VY_SPLIT_TRAIN")) : 1;
    if (train_split != 2 && net->conv_mode == VY_CONV_SPLIT_BF16X3_TRAIN && ci < c.t->dsplit.size() && c.t->dsplit[ci] >= 0) {
      a.w_split = net->dev_ws + c.t->dsplit_off + c.t->dsplit[ci];
      a.splitk_slabs = reinterpret_cast<float*>(net->dev_ws + net->sk_off + vy_net::al((size_t)VY_SK_FLAGS * sizeof(unsigned)));
      a.splitk_bytes = VY_SK_PARTIAL_BYTES;
    }
  }
  if (cv.stride == 1) {
    a.LH = ip.H;
    a.LW = ip.W;
    a.M = net->B * ip.H * ip.W;
    a.o_s = 1;
    a.o_oy = a.o_ox = 1;
    a.ntaps = cv.k * cv.k;
    for (int t = 0; t < a.ntaps; ++t) {
      a.tap_dy[t] = (signed char)(cv.k == 3 ? 1 - t / 3 : 0);
      a.tap_dx[t] = (signed char)(cv.k == 3 ? 1 - t % 3 : 0);
      a.tap_w[t] = (unsigned char)t;
    }
    out.a[out.n++] = a;
  } else {
    // stride 2, 3x3, pad 1: input pixel (2y'+py, 2x'+px) receives taps with (py+1-kh), (px+1-kw) even
    for (int py = 0; py < 2; ++py)
      for (int px = 0; px < 2; ++px) {
        ConvArgs q = a;
        q.LH = ip.H / 2;
        q.LW = ip.W / 2;
        q.M = net->B * q.LH * q.LW;
        q.o_s = 2;
        q.o_oy = 1 + py;
        q.o_ox = 1 + px;
        q.ntaps = 0;
        for (int kh = 0; kh < 3; ++kh) {
          if ((py + 1 - kh) & 1) continue;
          for (int kw = 0; kw < 3; ++kw) {
            if ((px + 1 - kw) & 1) continue;
            q.tap_dy[q.ntaps] = (signed char)((py + 1 - kh) / 2);
            q.tap_dx[q.ntaps] = (signed char)((px + 1 - kw) / 2);
            q.tap_w[q.ntaps] = (unsigned char)(kh * 3 + kw);
            ++q.ntaps;
          }
        }
        out.a[out.n++] = q;
      }
  }
  return out;
}

int launch_wgrad(const TrainCtx& c, size_t ci, const float* dzp, int dz_cs, int Ho, int Wo, hipStream_t ws) {
  vy_net* net = c.net;
  const ConvT& cv = net->convs[ci];
  const PlaneT& ip = net->planes[cv.in_plane];
  WgradArgs w;
  memset(&w, 0, sizeof w);
  w.dz = dzp;
  w.a = net->plane_ptr(cv.in_plane);
  w.slabs = c.slabs();
  w.zero = c.zero();
  w.B = net->B;
  w.Ho = Ho;
  w.Wo = Wo;
  w.M = net->B * Ho * Wo;
  w.z_cs = dz_cs;
  w.Cout = cv.cout;
  w.a_Hp = ip.H + 2;
  w.a_Wp = ip.W + 2;
  w.a_cs = ip.C;
  w.a_co = cv.in_co;
  w.stride = cv.stride;
  w.k = cv.k;
  w.Cin = cv.cin;
  w.splits = c.t->splits[ci];
  w.k_per_split = c.t->kps[ci];
  w.tab = reinterpret_cast<const uint2*>(net->dev_ws + c.t->tab_off[ci]);
  if (!g_labels_done) g_labels.note("wgrad", cv.name, w.M, w.Cout, (double)cv.k * cv.k * cv.cin);
  // conv mode VY_CONV_SPLIT_BF16X3_TRAIN: the split-fp32 weight-gradient kernel where it has the tile (Cout % 128 == 0)
  static const int wgrad_split = getenv("VY_SPLIT_WGRAD") ? atoi(getenv("VY_SPLIT_WGRAD")) : 1;
  if ((train_abl() & 8) || ((train_abl() & 32) && (long long)cv.k * cv.k * cv.cin <= 576 && cv.cout <= 128)) {
    // (bound measurements: no weight-gradient kernel at all / none for the early cells — N <= 128, K <= 576: stages.0.1 ... 0.5 —
    // whose output tile is mostly padding: what would a perfect kernel for them return to the step?)
  } else if (wgrad_split && net->conv_mode == VY_CONV_SPLIT_BF16X3_TRAIN && vy_wgrad_split_supported(w)) {
    HIP_TRY(vy_launch_wgrad_split(w, ws));
  } else {
    HIP_TRY(vy_launch_wgrad(w, ws));
  }
  HIP_TRY(vy_launch_slab_reduce(c.slabs(), w.splits, (long long)cv.cout * cv.k * cv.k * cv.cin,
                                c.grad_of(cv.p_weight), ws));
  return 0;
}

// the weight-gradient pixel tables depend on the planned shape only: built once after vy_net_bind_train
int build_wgrad_tables(const TrainCtx& c) {
  vy_net* net = c.net;
  for (size_t ci = 0; ci < net->convs.size(); ++ci) {
    const ConvT& cv = net->convs[ci];
    if (cv.is_stem) continue;
    const PlaneT& ip = net->planes[cv.in_plane];
    const int Ho = ip.H / cv.stride, Wo = ip.W / cv.stride;
    const long long M = (long long)net->B * Ho * Wo;
    const int z_cs = cv.p_gamma >= 0 ? c.t->z[ci].C : net->planes[cv.out_plane].C;
    if (M >= (1ll << 31) - 64) return fail(VY_ERR_UNSUPPORTED, "'%s': 2^31 output pixels or more in one batch", cv.name.c_str());
    // (offsets are relative to each split's first pixel: planes of 4 GiB and more are fine — 608x608 past batch 84)
    HIP_TRY(vy_launch_wgrad_table(net->dev_ws + c.t->tab_off[ci], (int)M, (int)vy_wgrad_table_entries(M), Ho, Wo, z_cs,
                                  ip.H + 2, ip.W + 2, ip.C, cv.stride, net->B, c.t->kps[ci], c.s));
  }
  c.t->tabs_built = true;
  return 0;
}

int backward_train(const TrainCtx& c, const float* x) {
  vy_net* net = c.net;
  const int B = net->B;
  if (!c.t->tabs_built)
    if (int rc = build_wgrad_tables(c)) return rc;
  // which channel ranges of each gradient plane already hold a contribution
  std::vector<std::vector<std::pair<int, int>>> touched(net->planes.size());
  auto covered = [&](int plane, int lo, int hi) -> int {  // 1 accumulate, 0 overwrite, -1 partial overlap
    for (auto& r : touched[plane])
      if (lo < r.second && r.first < hi) return (r.first <= lo && hi <= r.second) ? 1 : -1;
    return 0;
  };
  // the prediction planes' gradients were written by the loss kernel
  for (int i = 0; i < 3; ++i) touched[net->head_plane[i]].push_back({0, net->planes[net->head_plane[i]].C});
  // pending skip-connection gradients: block input plane view -> gradient view of the block output
  struct Skip {
    int plane, co;     // input view of the block (where the addend must land)
    int src_plane, src_co;
  };
  std::vector<Skip> skips;
  // gradient buckets: heads | stages.2 | stages.1 | stages.0 (contiguous parameter ranges)
  auto bucket_of = [&](const ConvT& cv) {
    if (cv.name.rfind("stages.2", 0) == 0) return 1;
    if (cv.name.rfind("stages.1", 0) == 0) return 2;
    if (cv.name.rfind("stages.0", 0) == 0) return 3;
    return 0;
  };
  auto join_side = [&]() -> int {
    if (!c.t->side) return 0;
    HIP_TRY(hipEventRecord(c.t->ev_side, c.t->side));
    HIP_TRY(hipStreamWaitEvent(c.s, c.t->ev_side, 0));
    return 0;
  };
  auto emit_bucket = [&](int bk) -> int {
    if (!c.t->gb_cb) return 0;
    if (int rc = join_side()) return rc;  // the bucket's weight gradients must be final
    int64_t lo = INT64_MAX, hi = 0;
    for (const ConvT& cv : net->convs) {
      if (bucket_of(cv) != bk) continue;
      const int ps[6] = {cv.p_weight, cv.p_gamma, cv.p_beta, cv.p_bias, -1, -1};
      for (int p : ps) {
        if (p < 0) continue;
        const vy_param_info& pi = net->params[p].info;
        if (pi.offset < lo) lo = pi.offset;
        const int64_t e = pi.offset + ((pi.size + 63) & ~(int64_t)63);
        if (e > hi) hi = e;
      }
    }
    if (hi <= lo) return 0;
    if (int rc = c.t->gb_cb(c.t->gb_user, lo, hi - lo)) return fail(VY_ERR_STATE, "gradient bucket callback failed (%d)", rc);
    return 0;
  };
  int cur_bucket = 0;
  for (int ci = (int)net->convs.size() - 1; ci >= 0; --ci) {
    const ConvT& cv = net->convs[ci];
    const int bk = bucket_of(cv);
    if (bk != cur_bucket) {
      if (int rc = emit_bucket(cur_bucket)) return rc;
      cur_bucket = bk;
    }
    const float* dzp;
    int dz_cs, dzH, dzW;
    if (cv.p_gamma < 0) {
      // prediction conv: dz = d(loss)/d(pred) as written by the loss kernel
      const PlaneT& pp = net->planes[cv.out_plane];
      dzp = c.gplane(cv.out_plane);
      dz_cs = pp.C;
      dzH = pp.H;
      dzW = pp.W;
      const int chunks = vy_colsum_chunks(B, pp.H, pp.W, kBwdChunk);
      HIP_TRY(vy_launch_colsum(dzp, B, pp.H, pp.W, pp.C, 0, cv.cout, kBwdChunk, c.partials(), c.s));
      HIP_TRY(vy_launch_reduce_partials(c.partials(), chunks, cv.cout, c.sums_local(), c.s));
      HIP_TRY(vy_launch_f64_to_f32(c.sums_local(), c.grad_of(cv.p_bias), cv.cout, c.s));
    } else {
      const ZPlane& zp = c.t->z[ci];
      const PlaneT& op = net->planes[cv.out_plane];
      if (covered(cv.out_plane, cv.out_co, cv.out_co + cv.cout) != 1)
        return fail(VY_ERR_STATE, "internal: gradient of '%s' output was never produced", cv.name.c_str());
      BnBwdArgs bb;
      memset(&bb, 0, sizeof bb);
      bb.g = c.gplane(cv.out_plane);
      bb.z = c.zplane(ci);
      bb.scale = net->dev_params + cv.scale_off;
      bb.shift = net->dev_params + cv.shift_off;
      bb.save_mean = c.save(ci);
      bb.save_invstd = c.save(ci) + ((cv.cout + 63) & ~63);
      bb.coef = c.coef();
      bb.partials = c.partials();
      bb.B = B;
      bb.H = zp.H;
      bb.W = zp.W;
      bb.C = cv.cout;
      bb.g_Hp = zp.H * cv.ups + 2;
      bb.g_Wp = zp.W * cv.ups + 2;
      bb.g_cs = op.C;
      bb.g_co = cv.out_co;
      bb.ups = cv.ups;
      bb.chunk = vy_bn_bwd_rows_per_chunk(B, zp.H, cv.cout);
      if (!(train_abl() & 1)) HIP_TRY(vy_launch_bn_bwd_reduce(bb, c.s));
      const bool exchange = sync_exchange(c.t) && is_sync_layer(cv);
      double count = (double)B * zp.H * zp.W;
      const double* use_sums = nullptr;
      if (exchange) {
        HIP_TRY(vy_launch_reduce_partials(c.partials(), vy_bn_bwd_chunks(bb), 2 * cv.cout, c.sums_local(), c.s));
        if (int rc = combine_sums(c, cv, 2 * cv.cout, &count, &use_sums)) return rc;
      }
      BnBwdFinalizeArgs f;
      memset(&f, 0, sizeof f);
      f.sums = use_sums;
      f.local_sums = c.sums_local();
      f.count = count;
      f.gamma = net->dev_params + net->params[cv.p_gamma].info.offset;
      f.save_invstd = bb.save_invstd;
      f.dgamma = c.grad_of(cv.p_gamma);
      f.dbeta = c.grad_of(cv.p_beta);
      f.coef = c.coef();
      f.C = cv.cout;
      if (exchange)
        HIP_TRY(vy_launch_bn_bwd_finalize(f, c.s));
      else if (!(train_abl() & (1 | 64)))  // (64: the finalize launch alone skipped)
        HIP_TRY(vy_launch_bn_bwd_reduce_finalize(c.partials(), vy_bn_bwd_chunks(bb), f, c.s));
      if (!(train_abl() & 4)) HIP_TRY(vy_launch_bn_bwd_apply(bb, c.s));
      dzp = c.zplane(ci);
      dz_cs = zp.C;
      dzH = zp.H;
      dzW = zp.W;
      if (cv.res_plane >= 0) skips.push_back({cv.res_plane, cv.res_co, cv.out_plane, cv.out_co});
    }
    // weight gradient
    if (cv.is_stem) {
      StemWgradArgs sw;
      sw.x = x;
      sw.dz = dzp;
      sw.partials = c.partials();
      sw.B = B;
      sw.H = net->H;
      sw.W = net->W;
      HIP_TRY(vy_launch_stem_wgrad(sw, c.s));
      HIP_TRY(vy_launch_reduce_partials(c.partials(), vy_stem_wgrad_blocks(B, net->H, net->W), 864,
                                        c.sums_local(), c.s));
      HIP_TRY(vy_launch_f64_to_f32(c.sums_local(), c.grad_of(cv.p_weight), 864, c.s));
      continue;  // no gradient w.r.t. the image
    }
    // dz is final on the main stream: the weight gradient (its own scratch: the slabs) goes to the side
    // stream and overlaps with this layer's dgrad and the next layers' BatchNorm kernels
    hipStream_t ws = c.t->side ? c.t->side : c.s;
    if (c.t->side) {
      HIP_TRY(hipEventRecord(c.t->ev_main, c.s));
      HIP_TRY(hipStreamWaitEvent(c.t->side, c.t->ev_main, 0));
    }
    if (int rc = launch_wgrad(c, (size_t)ci, dzp, dz_cs, dzH, dzW, ws)) return rc;
    // data gradient into the input view
    const int lo = cv.in_co, hi = cv.in_co + cv.cin;
    const int cov = covered(cv.in_plane, lo, hi);
    if (cov < 0) return fail(VY_ERR_STATE, "internal: partial gradient overlap at '%s'", cv.name.c_str());
    const float* addend = nullptr;
    int add_cs = 0, add_co = 0;
    int skip_i = -1;
    for (size_t k = 0; k < skips.size(); ++k)
      if (skips[k].plane == cv.in_plane && skips[k].co == cv.in_co) skip_i = (int)k;
    if (skip_i >= 0) {
      if (cov == 1) return fail(VY_ERR_STATE, "internal: skip + accumulate at '%s'", cv.name.c_str());
      addend = c.gplane(skips[skip_i].src_plane);
      add_cs = net->planes[skips[skip_i].src_plane].C;
      add_co = skips[skip_i].src_co;
      skips.erase(skips.begin() + skip_i);
    } else if (cov == 1) {
      addend = c.gplane(cv.in_plane);
      add_cs = net->planes[cv.in_plane].C;
      add_co = cv.in_co;
    }
    const BwdDgrad dg = make_dgrad(c, cv, dzp, dz_cs, dzH, dzW, addend, add_cs, add_co);
    for (int k = 0; k < dg.n; ++k) {
      if (!g_labels_done) g_labels.note("dgrad", cv.name, dg.a[k].M, dg.a[k].N, (double)dg.a[k].ntaps * dg.a[k].Kc);
      if (train_abl() & 16) continue;  // (bound measurement: no data-gradient kernel)
      if (int rc = launch_conv(dg.a[k], c.s)) return rc;
    }
    if (cov == 0) touched[cv.in_plane].push_back({lo, hi});
  }
  if (int rc = emit_bucket(cur_bucket)) return rc;
  if (int rc = join_side()) return rc;
  if (!skips.empty()) return fail(VY_ERR_STATE, "internal: unresolved skip gradient");
  if (!g_labels_done && g_labels.f) {
    g_labels.close_step();
    g_labels_done = true;
  }
  return 0;
}

}  // namespace

extern "C" {

// Training keeps the reference scripts' shapes: multiples of 32 (train_yolov3.py resizes to --data-shape, gluoncv's
// random shapes step by 32).  Inference takes any size (cropped upsample).
static int check_train_shape(int32_t height, int32_t width) {
  if (height % 32 || width % 32)
    return fail(VY_ERR_UNSUPPORTED, "training input %dx%d: height and width must be multiples of 32", height, width);
  return 0;
}

size_t vy_net_train_workspace_bytes(const vy_net* net, int32_t batch, int32_t height, int32_t width) {
  if (!net) return 0;
  if (check_train_shape(height, width)) return 0;
  if (vy_net_workspace_bytes(net, batch, height, width) == 0) return 0;
  return train_plan(const_cast<vy_net*>(net), batch, height, width, false);
}

int vy_net_bind_train(vy_net* net, void* dev_ws, size_t bytes, int32_t batch, int32_t height, int32_t width,
                      void* dev_grads, void* dev_momentum, void* stream) {
  if (!net || !dev_ws || !dev_grads || !dev_momentum) return fail(VY_ERR_INVALID, "null argument");
  if (int rc = check_train_shape(height, width)) return rc;
  if (vy_net_workspace_bytes(net, batch, height, width) == 0) return VY_ERR_INVALID;
  if (int rc = net->bind_cus(dev_ws)) return rc;
  const size_t need = train_plan(net, batch, height, width, false);
  if (bytes < need) return fail(VY_ERR_INVALID, "training workspace too small: %zu < %zu bytes", bytes, need);
  train_plan(net, batch, height, width, true);
  VyTrain* t = net->train;
  net->dev_ws = static_cast<unsigned char*>(dev_ws);
  net->ws_bytes = bytes;
  net->fold_uploaded = false;
  t->grads = static_cast<float*>(dev_grads);
  t->mom = static_cast<float*>(dev_momentum);
  t->forward_done = false;
  t->sdesc_uploaded = false;  // (the workspace is zeroed below)
  hipStream_t s = static_cast<hipStream_t>(stream);
  HIP_TRY(hipMemsetAsync(dev_ws, 0, need, s));
  net->sk_dirty = false;
  net->sk_ok = vy_sk_verify_topology(reinterpret_cast<unsigned*>(net->dev_ws + net->sk_off), s) != 0;
  static const int use_side = getenv("VY_TRAIN_SIDE_STREAM") ? atoi(getenv("VY_TRAIN_SIDE_STREAM")) : 1;
  if (use_side && !t->side) {
    // (The weight-gradient stream at the LOWEST queue priority was measured: +0.4 % on top of the raised issue priority of the
    // BatchNorm passes in a fresh process — and the whole training step 1.55x SLOWER, forward included, in a process that had
    // run the host-fed inference legs before (other streams alive: profiles/r06_ab_bn_prio.txt).  Default priority it stays.)
    HIP_TRY(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
    vy_bn_prio_init();
    HIP_TRY(hipEventCreateWithFlags(&t->ev_main, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&t->ev_side, hipEventDisableTiming));
  }
  return 0;
}

int vy_net_set_train_options(vy_net* net, float ignore_iou_thresh, int32_t label_smooth) {
  if (!net) return fail(VY_ERR_INVALID, "net is null");
  VyTrain* t = get_train(net);
  t->ignore_iou = ignore_iou_thresh;
  t->label_smooth = label_smooth ? 1 : 0;
  return 0;
}

static int train_forward_impl(vy_net* net, const float* x, const float* gt_boxes, int32_t M, const float* obj_t,
                         const float* centers_t, const float* scales_t, const float* weights_t,
                         const float* clas_t, float* losses, void* stream) {
  if (!net || !x || !obj_t || !centers_t || !scales_t || !weights_t || !clas_t || !losses || (M > 0 && !gt_boxes))
    return fail(VY_ERR_INVALID, "null argument");
  if (int rc = net->check_ready()) return rc;
  VyTrain* t = net->train;
  if (!t || !t->grads || t->B != net->B || t->H != net->H || t->W != net->W)
    return fail(VY_ERR_STATE, "training workspace not bound (vy_net_bind_train)");
  TrainCtx c{net, t, static_cast<hipStream_t>(stream)};
  if (int rc = forward_train(c, x)) return rc;
  const DetArgs d = net->det_args();
  LossArgs la;
  memset(&la, 0, sizeof la);
  int N = 0;
  for (int i = 0; i < 3; ++i) {
    la.head[i] = d.head[i];
    la.dpred[i] = c.gplane(net->head_plane[i]);
    N += 3 * d.head[i].H * d.head[i].W;
  }
  la.gt_boxes = gt_boxes;
  la.obj_t = obj_t;
  la.centers_t = centers_t;
  la.scales_t = scales_t;
  la.weights_t = weights_t;
  la.clas_t = clas_t;
  la.partials = reinterpret_cast<float*>(net->dev_ws + t->loss_part_off);
  la.B = net->B;
  la.C = net->num_class;
  la.M = M;
  la.N = N;
  la.ignore_iou_thresh = t->ignore_iou;
  la.label_smooth = t->label_smooth;
  HIP_TRY(vy_launch_loss(la, c.s));
  HIP_TRY(vy_launch_loss_reduce(la.partials, vy_loss_blocks_per_image(N), net->B, losses, c.s));
  t->forward_done = true;
  t->M = M;
  return 0;
}

int vy_net_train_forward(vy_net* net, const float* x, const float* gt_boxes, int32_t M, const float* obj_t,
                         const float* centers_t, const float* scales_t, const float* weights_t,
                         const float* clas_t, float* losses, void* stream) {
#ifdef VY_TRAIN_ABL_BUILD
  ++g_abl_forwards;
#endif
  if (net)
    if (int rc = net->sk_begin(static_cast<hipStream_t>(stream))) return rc;
  const int rc = train_forward_impl(net, x, gt_boxes, M, obj_t, centers_t, scales_t, weights_t, clas_t, losses, stream);
  return net ? net->sk_end(rc) : rc;
}

static int train_mode_forward_impl(vy_net* net, const float* x, float* box_preds, float* centers, float* scales,
                              float* objness, float* class_pred, void* stream) {
  if (!net || !x || !box_preds || !centers || !scales || !objness || !class_pred) return fail(VY_ERR_INVALID, "null argument");
  if (int rc = net->check_ready()) return rc;
  VyTrain* t = net->train;
  if (!t || !t->grads || t->B != net->B || t->H != net->H || t->W != net->W)
    return fail(VY_ERR_STATE, "training workspace not bound (vy_net_bind_train)");
  TrainCtx c{net, t, static_cast<hipStream_t>(stream)};
  if (int rc = forward_train(c, x)) return rc;
  t->forward_done = false;  // nothing was recorded: no backward may follow
  const DetArgs d = net->det_args();
  RawPredArgs ra;
  memset(&ra, 0, sizeof ra);
  int N = 0;
  for (int i = 0; i < 3; ++i) {
    ra.head[i] = d.head[i];
    N += 3 * d.head[i].H * d.head[i].W;
  }
  ra.box = box_preds;
  ra.centers = centers;
  ra.scales = scales;
  ra.objness = objness;
  ra.class_pred = class_pred;
  ra.B = net->B;
  ra.C = net->num_class;
  ra.N = N;
  HIP_TRY(vy_launch_raw_preds(ra, c.s));
  return 0;
}

int vy_net_train_mode_forward(vy_net* net, const float* x, float* box_preds, float* centers, float* scales,
                              float* objness, float* class_pred, void* stream) {
  if (net)
    if (int rc = net->sk_begin(static_cast<hipStream_t>(stream))) return rc;
  const int rc = train_mode_forward_impl(net, x, box_preds, centers, scales, objness, class_pred, stream);
  return net ? net->sk_end(rc) : rc;
}

static int train_backward_impl(vy_net* net, const float* x, void* stream) {
  if (!net || !x) return fail(VY_ERR_INVALID, "null argument");
  if (int rc = net->check_ready()) return rc;
  VyTrain* t = net->train;
  if (!t || !t->forward_done) return fail(VY_ERR_STATE, "vy_net_train_backward without a recorded forward");
  TrainCtx c{net, t, static_cast<hipStream_t>(stream)};
  t->forward_done = false;
  return backward_train(c, x);
}

int vy_net_train_backward(vy_net* net, const float* x, void* stream) {
  if (net)
    if (int rc = net->sk_begin(static_cast<hipStream_t>(stream))) return rc;
  const int rc = train_backward_impl(net, x, stream);
  return net ? net->sk_end(rc) : rc;
}

int vy_net_param_set_opt(vy_net* net, int32_t i, float lr_mult, float wd_mult, int32_t enabled) {
  if (!net || i < 0 || i >= (int32_t)net->params.size()) return fail(VY_ERR_INVALID, "bad argument");
  VyTrain* t = get_train(net);
  t->lr_mult[i] = lr_mult;
  t->wd_mult[i] = wd_mult;
  t->enabled[i] = enabled ? 1 : 0;
  // refresh the segment table in place if it is already planned
  int si = 0;
  for (int p = 0; p < (int)net->params.size(); ++p) {
    if (!net->params[p].info.trainable) continue;
    if (p == i && si < (int)t->segs.size()) {
      SgdSeg& sg = t->segs[si];
      const int en = enabled ? 1 : 0;
      if (sg.lr_mult != lr_mult || sg.wd_mult != wd_mult || sg.enabled != en) {  // re-upload only on a change
        sg.lr_mult = lr_mult;
        sg.wd_mult = wd_mult;
        sg.enabled = en;
        t->seg_uploaded = false;
      }
    }
    ++si;
  }
  return 0;
}

int vy_net_sgd_step(vy_net* net, float lr, float momentum, float wd, float rescale_grad, void* stream) {
  if (!net) return fail(VY_ERR_INVALID, "net is null");
  if (int rc = net->check_ready()) return rc;
  VyTrain* t = net->train;
  if (!t || !t->grads) return fail(VY_ERR_STATE, "training workspace not bound (vy_net_bind_train)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  SgdSeg* segs = reinterpret_cast<SgdSeg*>(net->dev_ws + t->seg_off);
  int32_t* chunks = reinterpret_cast<int32_t*>(net->dev_ws + t->chunk_off);
  if (!t->seg_uploaded) {
    HIP_TRY(hipMemcpyAsync(segs, t->segs.data(), t->segs.size() * sizeof(SgdSeg), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(chunks, t->chunk_seg.data(), t->chunk_seg.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // pageable host vectors; only when the table changed
    t->seg_uploaded = true;
  }
  HIP_TRY(vy_launch_sgd(net->dev_params, t->grads, t->mom, segs, chunks, (int)(t->chunk_seg.size() / 2), lr, momentum,
                        wd, rescale_grad, s));
  net->split_dirty = net->dsplit_dirty = net->wino_dirty = true;  // conv mode VY_CONV_SPLIT_BF16X3: the weight images are stale now
  return 0;
}

int vy_net_grad_get(vy_net* net, int32_t i, float* host_dst, void* stream) {
  if (!net || !host_dst || i < 0 || i >= (int32_t)net->params.size()) return fail(VY_ERR_INVALID, "bad argument");
  VyTrain* t = net->train;
  if (!t || !t->grads) return fail(VY_ERR_STATE, "training workspace not bound");
  const vy_param_info& pi = net->params[i].info;
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::vector<float> tmp((size_t)pi.size);
  HIP_TRY(hipMemcpyAsync(tmp.data(), t->grads + pi.offset, sizeof(float) * pi.size, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (pi.ndim == 4) {
    const int O = pi.shape[0], I = pi.shape[1], kk = pi.shape[2] * pi.shape[3];
    for (int o = 0; o < O; ++o)
      for (int ii = 0; ii < I; ++ii)
        for (int tt = 0; tt < kk; ++tt) host_dst[((size_t)o * I + ii) * kk + tt] = tmp[((size_t)o * kk + tt) * I + ii];
  } else {
    memcpy(host_dst, tmp.data(), sizeof(float) * pi.size);
  }
  return 0;
}

int vy_net_read_grad_activation(vy_net* net, const char* name, float* dst_dev, void* stream) {
  if (!net || !name || !dst_dev) return fail(VY_ERR_INVALID, "bad argument");
  if (int rc = net->check_ready()) return rc;
  VyTrain* t = net->train;
  if (!t || !t->grads) return fail(VY_ERR_STATE, "training workspace not bound");
  TrainCtx c{net, t, static_cast<hipStream_t>(stream)};
  for (const ConvT& cv : net->convs) {
    if (cv.name != name) continue;
    const PlaneT& p = net->planes[cv.out_plane];
    HIP_TRY(vy_launch_plane_to_nchw(c.gplane(cv.out_plane), net->B, p.H, p.W, p.C, cv.out_co, cv.cout, dst_dev, c.s));
    return 0;
  }
  return fail(VY_ERR_INVALID, "no cell named '%s'", name);
}

int vy_net_set_sync_bn(vy_net* net, int32_t world, vy_allreduce_cb cb, void* user) {
  if (!net || world < 1) return fail(VY_ERR_INVALID, "bad argument");
  if (world > 1 && !cb) return fail(VY_ERR_INVALID, "world > 1 needs an all-reduce callback");
  VyTrain* t = get_train(net);
  t->world = world;
  t->ar_cb = cb;
  t->ar_user = user;
  return 0;
}

int vy_net_set_grad_bucket_cb(vy_net* net, vy_grad_bucket_cb cb, void* user) {
  if (!net) return fail(VY_ERR_INVALID, "net is null");
  VyTrain* t = get_train(net);
  t->gb_cb = cb;
  t->gb_user = user;
  return 0;
}

}  // extern "C"
