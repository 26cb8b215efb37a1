// detect.hip — the detection tail: YOLOOutputV3 decode + contrib.box_nms + top post_nms rows.
//
// Replaces, without ever materialising the (B, N*C, 6) detection tensor the reference builds:
//   models/definitions/yolo/yolo3.py:158-197   (reshape/transpose, sigmoid/exp decode, x C tile,
//                                               class-major (B, C*HW*A, 6) layout)
//   models/definitions/yolo/yolo3.py:1195      (concat of the three scales, stride 32,16,8)
//   models/definitions/yolo/yolo3.py:1198-1206 (box_nms(valid 0.01, topk, per class) + slice)
//
// Candidate r of image b (the row index of the reference's detection tensor) is
//   r = cand_base[scale] + c*(H*W*A) + (y*W + x)*A + a
// and its sort key is (score, -r): box_nms orders by score descending and this implementation
// breaks score ties by ascending r (a stable sort of the reference's tensor).
//
// Pipeline per batch (all images in parallel), five launches:
//   1. hist (pass 0): decode every class score once from the 3 head planes, keep it ([B][C][anchors] fp32,
//      1.8 MB per image at 608/20 — a sixth of the reference's candidate tensor, no boxes, no ids) and
//      histogram it into 1024 linear score buckets;
//   2. select (pass 0): the bucket holding the k-th largest key, k = min(topk, #valid);
//   3. compact: one more sweep of the cached scores — candidates in HIGHER buckets are all wanted and go
//      straight to the entry list (box decoded), candidates IN the bucket are appended to a small per-image
//      list of keys (typically a few hundred of the 454 860);
//   4. refine: one workgroup per image finishes the radix select on that list (passes 1-3 refine the score
//      bits 10 at a time, passes 4-6 the inverted row index among candidates tying with the k-th score; a
//      pass is skipped once the threshold is exact) and appends the list members at or above the threshold
//      to the entries.  Only when the bucket overflows the list (degenerate inputs: every score equal) does
//      this kernel walk the whole score cache instead.  (Round 1 swept all cached scores in each of the six
//      refinement passes and once more to collect: 16 launches, 6x the bytes.)
//   5. sort_nms: one workgroup per image: rank sort by key, pair mask + word-wise greedy per-class IoU suppression,
//      compaction, write the first post_nms rows (-1 filler).
// HBM-bound integer/byte work: coalesced channel-contiguous reads, LDS histograms, no MFMA.
#include "kernels.h"
#include "../../include/vy_math.h"

namespace {

constexpr int kBins = 1024;
constexpr int kItemsPerThread = 4;
constexpr int kHistThreads = 256;
constexpr int kIdxBits = 30;

struct SelState {
  uint32_t Tb;            // selected linear bucket
  uint32_t Ts, smask;     // determined score bits / their mask
  uint32_t Ti, imask;     // determined inverted-index bits / their mask
  int32_t k_rem;          // rank still to resolve inside the current prefix
  int32_t k_eff;          // min(topk, nvalid)
  int32_t done;           // threshold exact: later passes are no-ops
  int32_t count;          // collect counter
  int32_t bucket_n;       // candidates in the selected bucket
  int32_t list_n;         // keys appended to the bucket list (== bucket_n unless it overflowed)
  int32_t pad[5];
};

constexpr int kListCap = 16384;  // bucket-list capacity per image (keys of 8 B)

struct Entry {
  uint32_t sbits, inv;
  float x1, y1, x2, y2;
  float cls;
  uint32_t pad;
};

struct Scratch {
  // [B] SelState | [B][kBins] hist | [B][VY_NMS_MAX_TOPK] Entry | [B][kListCap] key | [B][C][n_items] score
  SelState* st;
  uint32_t* hist;
  Entry* ent;
  unsigned long long* list;
  float* score;
};

__host__ __device__ inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__host__ __device__ inline Scratch carve(void* base, int B) {
  unsigned char* p = (unsigned char*)base;
  Scratch s;
  s.st = (SelState*)p;
  p += align256(sizeof(SelState) * (size_t)B);
  s.hist = (uint32_t*)p;
  p += align256(sizeof(uint32_t) * (size_t)B * kBins);
  s.ent = (Entry*)p;
  p += align256(sizeof(Entry) * (size_t)B * VY_NMS_MAX_TOPK);
  s.list = (unsigned long long*)p;
  p += align256(sizeof(unsigned long long) * (size_t)B * kListCap);
  s.score = (float*)p;
  return s;
}

__device__ __forceinline__ uint32_t score_bucket(float s) {
  int b = (int)(s * 1024.0f);
  return (uint32_t)(b > 1023 ? 1023 : b);
}

// Locate item `it` (anchor-level work item of one image) -> scale, cell, anchor, and a pointer to
// its 5+C raw predictions.  Items are ordered scale -> cell -> anchor like the candidate index.
struct Item {
  const float* p;   // raw predictions of this anchor: [x, y, w, h, obj, cls...]
  int scale, x, y, a;
  int cand0;        // candidate index of class 0 for this (cell, anchor)
  int cstride;      // candidate index stride between classes = H*W*A
};

__device__ __forceinline__ bool locate(const DetArgs& d, int b, int it, Item& o) {
  int s = 0;
#pragma unroll
  for (; s < 3; ++s) {
    const int n = d.head[s].H * d.head[s].W * 3;
    if (it < n) break;
    it -= n;
  }
  if (s == 3) return false;
  const HeadView& hv = d.head[s];
  const int a = it % 3;
  const int cell = it / 3;
  const int x = cell % hv.W, y = cell / hv.W;
  o.scale = s;
  o.x = x;
  o.y = y;
  o.a = a;
  o.cstride = hv.H * hv.W * 3;
  o.cand0 = hv.cand_base + it;
  o.p = hv.pred + ((long long)(b * (hv.H + 2) + y + 1) * (hv.W + 2) + x + 1) * hv.cs + hv.co + a * (5 + d.C);
  return true;
}

__device__ __forceinline__ bool prefix_match(const SelState& st, int pass, uint32_t bucket, uint32_t sbits,
                                             uint32_t inv) {
  if (pass == 0) return true;
  return bucket == st.Tb && (sbits & st.smask) == st.Ts && (inv & st.imask) == st.Ti;
}

__device__ __forceinline__ uint32_t pass_digit(int pass, uint32_t bucket, uint32_t sbits, uint32_t inv) {
  switch (pass) {
    case 0: return bucket;
    case 1: return (sbits >> 20) & 1023u;
    case 2: return (sbits >> 10) & 1023u;
    case 3: return sbits & 1023u;
    case 4: return (inv >> 20) & 1023u;
    case 5: return (inv >> 10) & 1023u;
    default: return inv & 1023u;
  }
}

// Pass 0: decode every class score, keep it, histogram it into the 1024 linear buckets.
// A block owns kPx consecutive pixels of ONE head of one image: their prediction vectors (kPx x cs floats,
// contiguous inside an image row of the padded plane) are copied to LDS with 16-B loads — every 128-B line of the
// head planes leaves HBM once — and decoded from there: objectness once per anchor, then one thread per
// (class, anchor), so that consecutive lanes write consecutive floats of the score cache.  (Round 1 / the first
// version of this round had one thread walk an anchor's 5 + C values in global memory: 4-byte loads at a 100-B
// stride, FETCH_SIZE 1.7 GB per batch for 0.2 GB of head planes.)
constexpr int kPxMax = 32;            // pixels per block (fewer when the prediction vectors are very wide)
constexpr int kTileFloats = 8192;     // LDS copy: kPx * cs floats; cs = 3 * (5 + C) rounded up to 32 (96 for VOC)

__host__ __device__ inline int hist_px(int cs) {
  const int p = kTileFloats / cs;
  return p > kPxMax ? kPxMax : (p < 1 ? 1 : p);
}
__host__ __device__ inline int hist_blocks_of(int hw, int cs) { return (hw + hist_px(cs) - 1) / hist_px(cs); }

__global__ __launch_bounds__(kHistThreads) void hist_kernel(const DetArgs d, void* scratch, int pass, int n_items) {
  const int b = blockIdx.y;
  Scratch sc = carve(scratch, d.B);
  __shared__ __attribute__((aligned(16))) float tile[kTileFloats];
  __shared__ float conf[kPxMax * 3];
  __shared__ uint32_t lh[kBins];
  // which head, which pixels
  int blk = blockIdx.x, s = 0, item0 = 0;
#pragma unroll
  for (; s < 3; ++s) {
    const int nb = hist_blocks_of(d.head[s].H * d.head[s].W, d.head[s].cs);
    if (blk < nb) break;
    blk -= nb;
    item0 += d.head[s].H * d.head[s].W * 3;
  }
  if (s == 3) return;
  const HeadView& hv = d.head[s];
  const int kPx = hist_px(hv.cs);
  const int hw = hv.H * hv.W, p0 = blk * kPx;
  const int npx = hw - p0 < kPx ? hw - p0 : kPx;
  const int cs4 = hv.cs >> 2;
  for (int i = threadIdx.x; i < kBins; i += kHistThreads) lh[i] = 0;
  for (int e = threadIdx.x; e < npx * cs4; e += kHistThreads) {
    const int px = e / cs4, q = e - px * cs4;
    const int p = p0 + px, x = p % hv.W, y = p / hv.W;
    const float4 v = *reinterpret_cast<const float4*>(
        hv.pred + ((long long)(b * (hv.H + 2) + y + 1) * (hv.W + 2) + x + 1) * hv.cs + hv.co + q * 4);
    *reinterpret_cast<float4*>(tile + px * hv.cs + q * 4) = v;
  }
  __syncthreads();
  const int P = 5 + d.C, nit = npx * 3;
  for (int i = threadIdx.x; i < nit; i += kHistThreads) conf[i] = vy_sigmoidf(tile[(i / 3) * hv.cs + (i % 3) * P + 4]);
  __syncthreads();
  float* cache = sc.score + (size_t)b * d.C * n_items + item0 + p0 * 3;
  const int cstride = hw * 3;
  for (int idx = threadIdx.x; idx < nit * d.C; idx += kHistThreads) {
    const int c = idx / nit, i = idx - c * nit;  // i = pixel * 3 + anchor inside the block: the candidate order
    const float sv = vy_sigmoidf(tile[(i / 3) * hv.cs + (i % 3) * P + 5 + c]) * conf[i];
    cache[(size_t)c * n_items + i] = sv;
    if (sv > d.valid_thresh) atomicAdd(&lh[score_bucket(sv)], 1u);
  }
  (void)cstride;
  __syncthreads();
  uint32_t* gh = sc.hist + (size_t)b * kBins;
  for (int i = threadIdx.x; i < kBins; i += kHistThreads) {
    const uint32_t v = lh[i];
    if (v) atomicAdd(&gh[i], v);
  }
}

// one block (kBins threads) per image: find the bin holding rank k_rem (counted from the top),
// fix that digit, zero the histogram for the next pass.
__global__ __launch_bounds__(kBins) void select_kernel(const DetArgs d, void* scratch, int pass) {
  const int b = blockIdx.x;
  Scratch sc = carve(scratch, d.B);
  SelState* st = sc.st + b;
  if (st->done) return;
  uint32_t* gh = sc.hist + (size_t)b * kBins;
  __shared__ uint32_t suf[kBins];  // suffix sums: suf[i] = sum_{j >= i} hist[j]
  const int t = threadIdx.x;
  const uint32_t mine = gh[t];
  gh[t] = 0;
  suf[t] = mine;
  __syncthreads();
  for (int off = 1; off < kBins; off <<= 1) {
    const uint32_t add = (t + off < kBins) ? suf[t + off] : 0u;
    __syncthreads();
    suf[t] += add;
    __syncthreads();
  }
  __shared__ int32_t k_rem_s;
  if (t == 0) {
    if (pass == 0) {
      const int nvalid = (int)suf[0];
      const int k_eff = nvalid < d.topk ? nvalid : d.topk;
      st->k_eff = k_eff;
      st->k_rem = k_eff;
      if (k_eff == 0) st->done = 1;
    }
    k_rem_s = st->k_rem;
  }
  __syncthreads();
  const int k_rem = k_rem_s;
  if (k_rem <= 0) return;
  const uint32_t above = (t + 1 < kBins) ? suf[t + 1] : 0u;  // candidates in strictly higher bins
  if (above < (uint32_t)k_rem && (uint32_t)k_rem <= above + mine) {
    // this is the bin
    const int rem = k_rem - (int)above;
    st->k_rem = rem;
    const uint32_t dgt = (uint32_t)t;
    switch (pass) {
      case 0: st->Tb = dgt; st->bucket_n = (int32_t)mine; break;
      case 1: st->Ts |= dgt << 20; st->smask |= 1023u << 20; break;
      case 2: st->Ts |= dgt << 10; st->smask |= 1023u << 10; break;
      case 3: st->Ts |= dgt; st->smask |= 1023u; break;
      case 4: st->Ti |= dgt << 20; st->imask |= 1023u << 20; break;
      case 5: st->Ti |= dgt << 10; st->imask |= 1023u << 10; break;
      default: st->Ti |= dgt; st->imask |= 1023u; break;
    }
    // every candidate of this bin is wanted: the threshold "prefix, rest zero" is already exact
    if ((uint32_t)rem == mine || pass == 6) st->done = 1;
  }
}

__device__ __forceinline__ void decode_box(const DetArgs& d, const Item& im, float& x1, float& y1, float& x2,
                                           float& y2) {
  const HeadView& hv = d.head[im.scale];
  // yolo3.py:172-177, same operation order (no contraction)
  const float cx = (vy_sigmoidf(im.p[0]) + (float)im.x) * hv.stride;
  const float cy = (vy_sigmoidf(im.p[1]) + (float)im.y) * hv.stride;
  const float w = vy_expf(im.p[2]) * hv.aw[im.a];
  const float h = vy_expf(im.p[3]) * hv.ah[im.a];
  const float hw = w / 2.0f, hh = h / 2.0f;
  x1 = cx - hw;
  y1 = cy - hh;
  x2 = cx + hw;
  y2 = cy + hh;
}

// candidate row r of image b -> its anchor item (for the box) and class
__device__ __forceinline__ void locate_cand(const DetArgs& d, int b, uint32_t r, Item& im, int& c) {
  int s = 2;
  if (r < (uint32_t)d.head[1].cand_base)
    s = 0;
  else if (r < (uint32_t)d.head[2].cand_base)
    s = 1;
  const HeadView& hv = d.head[s];
  const int n_s = hv.H * hv.W * 3;
  const int rel = (int)r - hv.cand_base;
  c = rel / n_s;
  const int it = rel - c * n_s;
  const int a = it % 3, cell = it / 3;
  im.scale = s;
  im.x = cell % hv.W;
  im.y = cell / hv.W;
  im.a = a;
  im.cstride = n_s;
  im.cand0 = hv.cand_base + it;
  im.p = hv.pred + ((long long)(b * (hv.H + 2) + im.y + 1) * (hv.W + 2) + im.x + 1) * hv.cs + hv.co + a * (5 + d.C);
}

__device__ __forceinline__ void put_entry(const DetArgs& d, Entry* ent, SelState* stp, const Item& im, int c,
                                          uint32_t sbits, uint32_t inv) {
  const int slot = atomicAdd(&stp->count, 1);
  if (slot >= VY_NMS_MAX_TOPK) return;
  Entry e;
  decode_box(d, im, e.x1, e.y1, e.x2, e.y2);
  e.sbits = sbits;
  e.inv = inv;
  e.cls = (float)c;
  e.pad = 0;
  ent[slot] = e;
}

// sweep of the cached scores after pass 0: higher buckets -> entries, the selected bucket -> key list.
// IPT anchor items per thread; blockIdx.z takes `cg_per_z` groups of 8 classes.  A big batch uses 4 items per thread and all
// classes in one block; a small one (fewer blocks than CUs that way: one frame at 608 x 608 is 23 blocks, 37 us of
// dependent round trips) one item and one class group per thread.
template <int IPT>
__global__ __launch_bounds__(kHistThreads) void compact_kernel(const DetArgs d, void* scratch, int n_items, int cg_per_z) {
  const int b = blockIdx.y;
  Scratch sc = carve(scratch, d.B);
  SelState* stp = sc.st + b;
  const uint32_t Tb = stp->Tb;
  if (stp->k_eff <= 0) return;
  const bool fits = stp->bucket_n <= kListCap;
  Entry* ent = sc.ent + (size_t)b * VY_NMS_MAX_TOPK;
  unsigned long long* list = sc.list + (size_t)b * kListCap;
  const int base = blockIdx.x * (kHistThreads * IPT);
  const int c_begin = blockIdx.z * cg_per_z * 8;
  const int c_end = c_begin + cg_per_z * 8 < d.C ? c_begin + cg_per_z * 8 : d.C;
#pragma unroll 1
  for (int q = 0; q < IPT; ++q) {
    const int it = base + q * kHistThreads + threadIdx.x;
    Item im;
    if (it >= n_items || !locate(d, b, it, im)) continue;
    const float* cache = sc.score + (size_t)b * d.C * n_items + it;
    // eight classes' scores in flight per round trip (one load per iteration made the sweep latency-bound at small
    // batches: 45 us for one frame's 1.8 MB)
    for (int c0 = c_begin; c0 < c_end; c0 += 8) {
      float s8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) s8[u] = (c0 + u < d.C) ? cache[(size_t)(c0 + u) * n_items] : 0.0f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float s = s8[u];
        const int c = c0 + u;
        if (c >= d.C || !(s > d.valid_thresh)) continue;
        const uint32_t bucket = score_bucket(s);
        if (bucket < Tb) continue;
        const uint32_t sbits = vy_f32_to_bits(s);
        const uint32_t inv = ((1u << kIdxBits) - 1u) - (uint32_t)(im.cand0 + c * im.cstride);
        if (bucket > Tb) {
          put_entry(d, ent, stp, im, c, sbits, inv);
        } else if (fits) {
          const int slot = atomicAdd(&stp->list_n, 1);
          if (slot < kListCap) list[slot] = ((unsigned long long)sbits << 32) | inv;
        }
      }
    }
  }
}

// one workgroup per image: passes 1-6 of the radix select on the bucket list (or, if it overflowed, on the whole
// score cache), then the list members at or above the exact threshold become entries
__global__ __launch_bounds__(kBins) void refine_kernel(const DetArgs d, void* scratch, int n_items) {
  const int b = blockIdx.x, t = threadIdx.x;
  Scratch sc = carve(scratch, d.B);
  SelState* stp = sc.st + b;
  __shared__ SelState st;
  __shared__ uint32_t hist[kBins];
  __shared__ uint32_t suf[kBins];
  if (t == 0) st = *stp;
  __syncthreads();
  if (st.k_eff <= 0) return;
  const bool from_list = st.bucket_n <= kListCap;
  const unsigned long long* list = sc.list + (size_t)b * kListCap;
  const float* cache = sc.score + (size_t)b * d.C * n_items;
  const int n = from_list ? st.bucket_n : d.C * n_items;
  const int n0 = d.head[0].H * d.head[0].W * 3, n1 = d.head[1].H * d.head[1].W * 3;
  // element idx of the source -> (in the bucket?, sbits, inv)
  auto fetch = [&](int idx, uint32_t& sbits, uint32_t& inv) -> bool {
    if (from_list) {
      const unsigned long long k = list[idx];
      sbits = (uint32_t)(k >> 32);
      inv = (uint32_t)k;
      return true;
    }
    const float s = cache[idx];
    if (!(s > d.valid_thresh) || score_bucket(s) != st.Tb) return false;
    const int c = idx / n_items, it = idx - c * n_items;
    uint32_t r;
    if (it < n0)
      r = (uint32_t)(d.head[0].cand_base + c * n0 + it);
    else if (it < n0 + n1)
      r = (uint32_t)(d.head[1].cand_base + c * n1 + (it - n0));
    else
      r = (uint32_t)(d.head[2].cand_base + c * (n_items - n0 - n1) + (it - n0 - n1));
    sbits = vy_f32_to_bits(s);
    inv = ((1u << kIdxBits) - 1u) - r;
    return true;
  };
  for (int pass = 1; pass < 7; ++pass) {
    if (st.done) break;  // uniform
    hist[t] = 0;
    __syncthreads();
    for (int idx = t; idx < n; idx += kBins) {
      uint32_t sbits, inv;
      if (!fetch(idx, sbits, inv)) continue;
      if ((sbits & st.smask) != st.Ts || (inv & st.imask) != st.Ti) continue;
      atomicAdd(&hist[pass_digit(pass, st.Tb, sbits, inv)], 1u);
    }
    __syncthreads();
    const uint32_t mine = hist[t];
    suf[t] = mine;
    __syncthreads();
    for (int off = 1; off < kBins; off <<= 1) {
      const uint32_t add = (t + off < kBins) ? suf[t + off] : 0u;
      __syncthreads();
      suf[t] += add;
      __syncthreads();
    }
    const int k_rem = st.k_rem;
    const uint32_t above = (t + 1 < kBins) ? suf[t + 1] : 0u;
    __syncthreads();
    if (k_rem > 0 && above < (uint32_t)k_rem && (uint32_t)k_rem <= above + mine) {
      st.k_rem = k_rem - (int)above;
      const uint32_t dgt = (uint32_t)t;
      switch (pass) {
        case 1: st.Ts |= dgt << 20; st.smask |= 1023u << 20; break;
        case 2: st.Ts |= dgt << 10; st.smask |= 1023u << 10; break;
        case 3: st.Ts |= dgt; st.smask |= 1023u; break;
        case 4: st.Ti |= dgt << 20; st.imask |= 1023u << 20; break;
        case 5: st.Ti |= dgt << 10; st.imask |= 1023u << 10; break;
        default: st.Ti |= dgt; st.imask |= 1023u; break;
      }
      if ((uint32_t)(k_rem - (int)above) == mine || pass == 6) st.done = 1;
    }
    __syncthreads();
  }
  // the bucket's members at or above the threshold
  Entry* ent = sc.ent + (size_t)b * VY_NMS_MAX_TOPK;
  for (int idx = t; idx < n; idx += kBins) {
    uint32_t sbits, inv;
    if (!fetch(idx, sbits, inv)) continue;
    if (!(sbits > st.Ts || (sbits == st.Ts && inv >= st.Ti))) continue;
    Item im;
    int c;
    locate_cand(d, b, ((1u << kIdxBits) - 1u) - inv, im, c);
    put_entry(d, ent, stp, im, c, sbits, inv);
  }
}

constexpr int kNmsThreads = 1024;

#ifdef VY_NMS_TRACE  // tools/nms_latency.py --trace: phase stamps of image 0's workgroup (100 MHz realtime counter)
__device__ unsigned long long vy_nms_trace_buf[16];
#define NMS_STAMP(n) do { if (t == 0 && b == 0) vy_nms_trace_buf[n] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int vy_debug_nms_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vy_nms_trace_buf), sizeof(vy_nms_trace_buf));
}
#else
#define NMS_STAMP(n) do { } while (0)
#endif

// one workgroup per image: sort the <= topk collected entries by key (descending), greedy
// per-class suppression in that order, compact, write the first `rows` rows.
// (round 4: at batch 1 this workgroup is 5 % of the frame's latency — 91 us for 400 candidates of one class, measured by
// phase with -DVY_NMS_TRACE: bitonic sort 12, pair mask 53, greedy pass 23.  Now: 1024 threads; rank sort without
// barriers; the pair mask enumerated over the words that can hold a later candidate only; the greedy pass word by
// word — a 32-step scalar chain inside a word, its survivors applied to all later words at once.)
__global__ __launch_bounds__(kNmsThreads) void sort_nms_kernel(const DetArgs d, void* scratch, int rows, float* ids,
                                                               float* scores, float* bboxes, int32_t* keep_idx) {
  static_assert(kNmsThreads == VY_NMS_MAX_TOPK, "one candidate per thread");
  const int b = blockIdx.x;
  Scratch sc = carve(scratch, d.B);
  const SelState st = sc.st[b];
  const Entry* ent = sc.ent + (size_t)b * VY_NMS_MAX_TOPK;
  __shared__ __attribute__((aligned(16))) unsigned long long key[VY_NMS_MAX_TOPK];
  __shared__ float bx1[VY_NMS_MAX_TOPK], by1[VY_NMS_MAX_TOPK], bx2[VY_NMS_MAX_TOPK], by2[VY_NMS_MAX_TOPK];
  __shared__ float bcls[VY_NMS_MAX_TOPK];
  __shared__ uint8_t alive[VY_NMS_MAX_TOPK];
  __shared__ int pos[VY_NMS_MAX_TOPK];
  constexpr int kMaskRows = 416;  // pairwise suppression bits for up to 416 candidates (13 words each): 21 KiB
  __shared__ uint32_t mask[kMaskRows][13];
  const int t = threadIdx.x;
  NMS_STAMP(0);
  int k = st.k_eff;
  if (k > VY_NMS_MAX_TOPK) k = VY_NMS_MAX_TOPK;
  Entry e = {};
  unsigned long long my_key = 0ull;  // key 0 (slots at and past k) is below every valid key: valid keys have sbits > 0
  if (t < k) {
    e = ent[t];
    my_key = ((unsigned long long)e.sbits << 32) | e.inv;
  }
  key[t] = my_key;
  pos[t] = 0;
  __syncthreads();
  NMS_STAMP(1);
  // Rank sort, descending: the keys are distinct (their low half is the candidate index), so a candidate's place is the
  // number of larger keys.  All 1024 threads count: candidate t & (kp - 1), a 1 / G slice of the keys each, four keys
  // per step (two 16-byte LDS reads of addresses the whole wave shares: broadcasts); no barrier until the ranks are
  // complete.  (The bitonic network this replaces: 45 barrier-separated passes for 400 candidates, 12 us.)
  {
    int kp = 1;
    while (kp < k) kp <<= 1;
    const int G = kNmsThreads / kp, i = t & (kp - 1), part = t / kp;
    const int len = (((k + G - 1) / G) + 3) & ~3;
    if (i < k) {
      typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
      const unsigned long long my = key[i];
      const int k4 = (k + 3) & ~3;  // <= 1024: slots k .. k4 hold 0
      int j = part * len, cnt = 0;
      const int j_end = j + len < k4 ? j + len : k4;
      for (; j < j_end; j += 4) {
        const u64x2 a = *reinterpret_cast<const u64x2*>(&key[j]), c = *reinterpret_cast<const u64x2*>(&key[j + 2]);
        cnt += (a[0] > my ? 1 : 0) + (a[1] > my ? 1 : 0) + (c[0] > my ? 1 : 0) + (c[1] > my ? 1 : 0);
      }
      if (cnt) atomicAdd(&pos[i], cnt);
    }
  }
  __syncthreads();
  NMS_STAMP(2);
  if (t < k) {
    const int r = pos[t];
    key[r] = my_key;  // every key[] was read before the barrier above
    bx1[r] = e.x1;
    by1[r] = e.y1;
    bx2[r] = e.x2;
    by2[r] = e.y2;
    bcls[r] = e.cls;
    alive[r] = 1;
  }
  __syncthreads();
  NMS_STAMP(3);
  if (d.do_nms && k <= kMaskRows) {
    // the usual case (topk 400): all pairwise "i would suppress j" bits in parallel, then the greedy pass, whose only
    // serial dependency is the alive mask.  Word w of row i is needed only if it can hold a j > i: i < 32 (w + 1).
    // Items are enumerated word by word, rows innermost: a wave works on consecutive rows of one word and reads the
    // same candidate j in every step (LDS broadcasts), its lanes differ only in the row they hold in registers.
    const int words = (k + 31) >> 5;
    const int full = k >> 5;  // words w with 32 (w + 1) <= k
    const int total = 16 * full * (full + 1) + (words > full ? k : 0);
    for (int idx = t; idx < total; idx += kNmsThreads) {
      int w = 0, i = idx;
      for (;;) {
        const int n_w = 32 * (w + 1) < k ? 32 * (w + 1) : k;
        if (i < n_w) break;
        i -= n_w;
        ++w;
      }
      const float ac = bcls[i], ax1 = bx1[i], ay1 = by1[i], ax2 = bx2[i], ay2 = by2[i];
      uint32_t m = 0;
#pragma unroll 4
      for (int bit = 0; bit < 32; ++bit) {
        const int j = 32 * w + bit;
        if (j > i && j < k && bcls[j] == ac &&
            vy_box_iou(ax1, ay1, ax2, ay2, bx1[j], by1[j], bx2[j], by2[j]) > d.nms_thresh)
          m |= 1u << bit;
      }
      mask[i][w] = m;
    }
    __syncthreads();
    NMS_STAMP(4);
    if (t < 64) {
      // alive bits: word l in lane l.  Words are settled in order.  Word wb, once every earlier survivor has been applied
      // to it: a 32-step chain over its own candidates on the scalar unit (lane l holds the diagonal word of row 32 wb + l,
      // v_readlane with a constant lane), then the rows of its survivors are or-ed into all later words at once
      // (32 independent LDS reads per lane, selected by an all-ones / all-zeros scalar).  Rows at or past k are never
      // selected (their alive bits start as 0); mask[i][w] with w < i / 32 is never read.
      uint32_t aw = 0;
      if (t < words) aw = (32 * t + 32 <= k) ? 0xffffffffu : ((1u << (k - 32 * t)) - 1u);
      const int tw = t < words ? t : 0;
      for (int wb = 0; wb < words; ++wb) {
        uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)aw, wb);
        if (cur == 0u) continue;  // uniform
        const int di = 32 * wb + (t & 31);
        const uint32_t diag = di < k ? mask[di][wb] : 0u;
        if (__builtin_amdgcn_ballot_w64(diag != 0u) != 0ull) {  // (no pair inside this word overlaps: nothing to settle)
#pragma unroll
          for (int bit = 0; bit < 32; ++bit) {
            const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)diag, bit);
            const uint32_t sel = 0u - ((cur >> bit) & 1u);
            cur &= ~(row & sel);
          }
        }
        uint32_t acc0 = 0u, acc1 = 0u;
        if (wb + 1 < words) {
#pragma unroll
          for (int bit = 0; bit < 32; bit += 2) {
            acc0 |= mask[32 * wb + bit][tw] & (0u - ((cur >> bit) & 1u));
            acc1 |= mask[32 * wb + bit + 1][tw] & (0u - ((cur >> (bit + 1)) & 1u));
          }
        }
        if (t == wb) aw = cur;
        else if (t > wb) aw &= ~(acc0 | acc1);
      }
      if (t < words)
        for (int bit = 0; bit < 32 && 32 * t + bit < k; ++bit) alive[32 * t + bit] = (uint8_t)((aw >> bit) & 1u);
    }
    __syncthreads();
  } else if (d.do_nms) {
    for (int i = 0; i < k; ++i) {
      if (!alive[i]) continue;  // uniform: alive[] is only written before the barrier below
      const float ax1 = bx1[i], ay1 = by1[i], ax2 = bx2[i], ay2 = by2[i], ac = bcls[i];
      for (int j = i + 1 + t; j < k; j += kNmsThreads) {
        if (alive[j] && bcls[j] == ac) {
          const float iou = vy_box_iou(ax1, ay1, ax2, ay2, bx1[j], by1[j], bx2[j], by2[j]);
          if (iou > d.nms_thresh) alive[j] = 0;
        }
      }
      __syncthreads();
    }
  }
  // compaction: inclusive scan of alive[] (k <= 1024, one element per thread)
  NMS_STAMP(5);
  __syncthreads();  // (the rank counters in pos[] were last read before the barriers above)
  pos[t] = (t < k && alive[t]) ? 1 : 0;
  __syncthreads();
  for (int off = 1; off < VY_NMS_MAX_TOPK; off <<= 1) {
    const int v = t >= off ? pos[t - off] : 0;
    __syncthreads();
    pos[t] += v;
    __syncthreads();
  }
  const int n_keep = pos[VY_NMS_MAX_TOPK - 1];
  NMS_STAMP(6);
  for (int i = t; i < k; i += kNmsThreads) {
    if (!alive[i]) continue;
    const int r = pos[i] - 1;
    if (r >= rows) continue;
    const size_t o = (size_t)b * rows + r;
    ids[o] = bcls[i];
    scores[o] = vy_bits_to_f32((uint32_t)(key[i] >> 32));
    bboxes[o * 4 + 0] = bx1[i];
    bboxes[o * 4 + 1] = by1[i];
    bboxes[o * 4 + 2] = bx2[i];
    bboxes[o * 4 + 3] = by2[i];
    if (keep_idx) keep_idx[o] = (int32_t)(((1u << kIdxBits) - 1u) - (uint32_t)(key[i] & 0xffffffffull));
  }
  for (int r = n_keep + t; r < rows; r += kNmsThreads) {
    const size_t o = (size_t)b * rows + r;
    ids[o] = -1.0f;
    scores[o] = -1.0f;
    bboxes[o * 4 + 0] = -1.0f;
    bboxes[o * 4 + 1] = -1.0f;
    bboxes[o * 4 + 2] = -1.0f;
    bboxes[o * 4 + 3] = -1.0f;
    if (keep_idx) keep_idx[o] = -1;
  }
  NMS_STAMP(7);
}


// ---------------------------------------------------------------------------------------------------
// nms_topk <= 0 ("use -1 to disable so that every detection is used for NMS", yolo3.py:1208-1228): box_nms then
// sorts EVERY valid candidate (up to N*C per image) and suppresses greedily; the caller keeps the first
// post_nms survivors.  A survivor only depends on the survivors before it, so the sorted list can be consumed
// in chunks: one workgroup per image repeatedly (1) radix-selects the next <= 1024 largest keys below the
// previous chunk's threshold from the cached class scores, (2) decodes and bitonic-sorts them, (3) suppresses
// them against the rows kept so far and among themselves, (4) appends the survivors to the output — until
// post_nms rows are kept or the valid candidates are exhausted.  Typical inputs finish in one or two chunks; the
// worst case (everything suppressed) walks all N*C candidates, like the reference's unbounded O(n^2) loop.
constexpr int kAllThreads = 1024;

// The same loop serves nms_topk > VY_NMS_MAX_TOPK (`topk_cap` > 0): box_nms keeps the topk best valid candidates and
// suppresses among those, so the chunks stop once topk_cap candidates have been consumed (the last chunk is
// shortened to what is left of the cap).
__global__ __launch_bounds__(kAllThreads) void nms_all_kernel(const DetArgs d, void* scratch, int n_items, int rows,
                                                              int topk_cap, float* ids, float* scores, float* bboxes,
                                                              int32_t* keep_idx) {
  const int b = blockIdx.x, t = threadIdx.x;
  Scratch sc = carve(scratch, d.B);
  const float* cache = sc.score + (size_t)b * d.C * n_items;
  __shared__ uint32_t hist[kBins];
  __shared__ unsigned long long key[VY_NMS_MAX_TOPK];
  __shared__ uint16_t perm[VY_NMS_MAX_TOPK];
  __shared__ float bx1[VY_NMS_MAX_TOPK], by1[VY_NMS_MAX_TOPK], bx2[VY_NMS_MAX_TOPK], by2[VY_NMS_MAX_TOPK];
  __shared__ float bcls[VY_NMS_MAX_TOPK];
  __shared__ uint8_t alive[VY_NMS_MAX_TOPK];
  __shared__ int pos[VY_NMS_MAX_TOPK];
  __shared__ float kx1[VY_NMS_MAX_TOPK], ky1[VY_NMS_MAX_TOPK], kx2[VY_NMS_MAX_TOPK], ky2[VY_NMS_MAX_TOPK];
  __shared__ float kcls[VY_NMS_MAX_TOPK];
  __shared__ SelState st;
  __shared__ int n_kept, n_taken;
  const int n0 = d.head[0].H * d.head[0].W * 3, n1 = d.head[1].H * d.head[1].W * 3;
  const int total = d.C * n_items;
  const bool kept_in_lds = rows <= VY_NMS_MAX_TOPK;
  // candidate row of cached score idx = c*n_items + item
  auto cand_of = [&](int idx, int& c, int& it) -> uint32_t {
    c = idx / n_items;
    it = idx - c * n_items;
    if (it < n0) return (uint32_t)(d.head[0].cand_base + c * n0 + it);
    if (it < n0 + n1) return (uint32_t)(d.head[1].cand_base + c * n1 + (it - n0));
    const int n2 = n_items - n0 - n1;
    return (uint32_t)(d.head[2].cand_base + c * n2 + (it - n0 - n1));
  };
  if (t == 0) n_kept = 0;
  // exclusive upper bound on the key of the candidates still to be consumed (none yet)
  uint32_t bound_s = 0xffffffffu, bound_i = 0xffffffffu;
  bool first = true;
  int consumed = 0;  // candidates taken by earlier chunks (uniform)
  __syncthreads();
  while (true) {
    const int k_limit = (topk_cap > 0 && topk_cap - consumed < VY_NMS_MAX_TOPK) ? topk_cap - consumed : VY_NMS_MAX_TOPK;
    if (t == 0) {
      st.Tb = st.Ts = st.smask = st.Ti = st.imask = 0;
      st.k_rem = st.k_eff = st.done = st.count = 0;
      n_taken = 0;
    }
    __syncthreads();
    // ---- radix select of the min(1024, remaining)-th largest remaining key
    for (int pass = 0; pass < 7; ++pass) {
      if (st.done) break;  // uniform: st is only written between barriers
      hist[t] = 0;
      __syncthreads();
      for (int idx = t; idx < total; idx += kAllThreads) {
        const float s = cache[idx];
        if (!(s > d.valid_thresh)) continue;
        int c, it;
        const uint32_t sbits = vy_f32_to_bits(s);
        const uint32_t inv = ((1u << kIdxBits) - 1u) - cand_of(idx, c, it);
        if (!first && !(sbits < bound_s || (sbits == bound_s && inv < bound_i))) continue;
        const uint32_t bucket = score_bucket(s);
        if (!prefix_match(st, pass, bucket, sbits, inv)) continue;
        atomicAdd(&hist[pass_digit(pass, bucket, sbits, inv)], 1u);
      }
      __syncthreads();
      const uint32_t mine = hist[t];
      pos[t] = (int)mine;  // suffix sums in pos[]
      __syncthreads();
      for (int off = 1; off < kBins; off <<= 1) {
        const int add = (t + off < kBins) ? pos[t + off] : 0;
        __syncthreads();
        pos[t] += add;
        __syncthreads();
      }
      if (t == 0 && pass == 0) {
        const int nvalid = pos[0];
        st.k_eff = nvalid < k_limit ? nvalid : k_limit;
        st.k_rem = st.k_eff;
        if (st.k_eff == 0) st.done = 1;
      }
      __syncthreads();
      const int k_rem = st.k_rem;
      const uint32_t above = (t + 1 < kBins) ? (uint32_t)pos[t + 1] : 0u;
      __syncthreads();
      if (k_rem > 0 && above < (uint32_t)k_rem && (uint32_t)k_rem <= above + mine) {
        const int rem = k_rem - (int)above;
        st.k_rem = rem;
        const uint32_t dgt = (uint32_t)t;
        switch (pass) {
          case 0: st.Tb = dgt; break;
          case 1: st.Ts |= dgt << 20; st.smask |= 1023u << 20; break;
          case 2: st.Ts |= dgt << 10; st.smask |= 1023u << 10; break;
          case 3: st.Ts |= dgt; st.smask |= 1023u; break;
          case 4: st.Ti |= dgt << 20; st.imask |= 1023u << 20; break;
          case 5: st.Ti |= dgt << 10; st.imask |= 1023u << 10; break;
          default: st.Ti |= dgt; st.imask |= 1023u; break;
        }
        if ((uint32_t)rem == mine || pass == 6) st.done = 1;
      }
      __syncthreads();
    }
    const int k = st.k_eff;
    if (k == 0) break;  // no valid candidate left
    // ---- collect + decode the chunk
    for (int idx = t; idx < total; idx += kAllThreads) {
      const float s = cache[idx];
      if (!(s > d.valid_thresh)) continue;
      int c, it;
      const uint32_t sbits = vy_f32_to_bits(s);
      const uint32_t inv = ((1u << kIdxBits) - 1u) - cand_of(idx, c, it);
      if (!first && !(sbits < bound_s || (sbits == bound_s && inv < bound_i))) continue;
      const uint32_t bucket = score_bucket(s);
      const bool take = bucket > st.Tb || (bucket == st.Tb && (sbits > st.Ts || (sbits == st.Ts && inv >= st.Ti)));
      if (!take) continue;
      const int slot = atomicAdd(&n_taken, 1);
      if (slot >= VY_NMS_MAX_TOPK) continue;  // cannot happen: the threshold is exact
      Item im;
      locate(d, b, it, im);
      decode_box(d, im, bx1[slot], by1[slot], bx2[slot], by2[slot]);
      bcls[slot] = (float)c;
      key[slot] = ((unsigned long long)sbits << 32) | inv;
      perm[slot] = (uint16_t)slot;
    }
    __syncthreads();
    int kp = 1;
    while (kp < k) kp <<= 1;
    for (int i = k + t; i < kp; i += kAllThreads) {
      key[i] = 0ull;
      perm[i] = 0;
    }
    __syncthreads();
    for (int size = 2; size <= kp; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = t; i < (kp >> 1); i += kAllThreads) {
          const int lo = ((i / stride) * (stride << 1)) + (i % stride);
          const int hi = lo + stride;
          const bool desc = ((lo & size) == 0);
          const unsigned long long a = key[lo], c = key[hi];
          if ((a < c) == desc) {
            key[lo] = c;
            key[hi] = a;
            const uint16_t pa = perm[lo];
            perm[lo] = perm[hi];
            perm[hi] = pa;
          }
        }
        __syncthreads();
      }
    }
    // ---- suppression: first by the rows kept in earlier chunks, then greedily inside the chunk.  The kept rows live
    // in LDS while the output has at most VY_NMS_MAX_TOPK rows; a longer output (nms_topk > 1024 or <= 0 together with
    // post_nms <= 0 or > 1024) IS the list of kept rows, so they are read back from it (written by this workgroup,
    // fenced at workgroup scope below)
    const int nk = n_kept;
    if (t < k) {
      const int e = perm[t];
      bool ok = true;
      if (kept_in_lds) {
        for (int j = 0; j < nk && ok; ++j)
          if (kcls[j] == bcls[e] && vy_box_iou(kx1[j], ky1[j], kx2[j], ky2[j], bx1[e], by1[e], bx2[e], by2[e]) > d.nms_thresh)
            ok = false;
      } else {
        const float* kb = bboxes + (size_t)b * rows * 4;
        const float* kc = ids + (size_t)b * rows;
        for (int j = 0; j < nk && ok; ++j)
          if (kc[j] == bcls[e] &&
              vy_box_iou(kb[j * 4 + 0], kb[j * 4 + 1], kb[j * 4 + 2], kb[j * 4 + 3], bx1[e], by1[e], bx2[e], by2[e]) > d.nms_thresh)
            ok = false;
      }
      alive[t] = ok ? 1 : 0;
    }
    __syncthreads();
    for (int i = 0; i < k; ++i) {
      if (!alive[i]) continue;  // uniform
      const int ei = perm[i];
      const float ax1 = bx1[ei], ay1 = by1[ei], ax2 = bx2[ei], ay2 = by2[ei], ac = bcls[ei];
      for (int j = i + 1 + t; j < k; j += kAllThreads) {
        const int ej = perm[j];
        if (alive[j] && bcls[ej] == ac && vy_box_iou(ax1, ay1, ax2, ay2, bx1[ej], by1[ej], bx2[ej], by2[ej]) > d.nms_thresh)
          alive[j] = 0;
      }
      __syncthreads();
    }
    // ---- append the survivors in order
    pos[t] = (t < k && alive[t]) ? 1 : 0;
    __syncthreads();
    for (int off = 1; off < VY_NMS_MAX_TOPK; off <<= 1) {
      const int v = t >= off ? pos[t - off] : 0;
      __syncthreads();
      pos[t] += v;
      __syncthreads();
    }
    const int add = pos[VY_NMS_MAX_TOPK - 1];
    if (t < k && alive[t]) {
      const int r = nk + pos[t] - 1;
      if (r < rows) {
        const int e = perm[t];
        if (kept_in_lds) {
          kx1[r] = bx1[e];
          ky1[r] = by1[e];
          kx2[r] = bx2[e];
          ky2[r] = by2[e];
          kcls[r] = bcls[e];
        }
        const size_t o = (size_t)b * rows + r;
        ids[o] = bcls[e];
        scores[o] = vy_bits_to_f32((uint32_t)(key[t] >> 32));
        bboxes[o * 4 + 0] = bx1[e];
        bboxes[o * 4 + 1] = by1[e];
        bboxes[o * 4 + 2] = bx2[e];
        bboxes[o * 4 + 3] = by2[e];
        if (keep_idx) keep_idx[o] = (int32_t)(((1u << kIdxBits) - 1u) - (uint32_t)(key[t] & 0xffffffffull));
      }
    }
    // next chunk: everything strictly below this chunk's smallest key
    const unsigned long long last = key[k - 1];
    if (!kept_in_lds) __threadfence_block();  // the rows just written are the next chunk's kept list
    __syncthreads();
    if (t == 0) n_kept = (nk + add < rows) ? nk + add : rows;
    __syncthreads();
    consumed += k;
    // enough rows, or the candidates are exhausted, or the topk best have all been through
    if (n_kept >= rows || k < k_limit || (topk_cap > 0 && consumed >= topk_cap)) break;
    bound_s = (uint32_t)(last >> 32);
    bound_i = (uint32_t)(last & 0xffffffffull);
    first = false;
  }
  __syncthreads();
  for (int r = n_kept + t; r < rows; r += kAllThreads) {
    const size_t o = (size_t)b * rows + r;
    ids[o] = -1.0f;
    scores[o] = -1.0f;
    bboxes[o * 4 + 0] = -1.0f;
    bboxes[o * 4 + 1] = -1.0f;
    bboxes[o * 4 + 2] = -1.0f;
    bboxes[o * 4 + 3] = -1.0f;
    if (keep_idx) keep_idx[o] = -1;
  }
}

// nms_thresh outside (0,1): the reference returns the un-suppressed detection tensor itself (yolo3.py:1195-1206
// with the box_nms branch skipped): (B, N*C, 6) rows [id, score, x1, y1, x2, y2] in class-major order per scale.
// One thread per anchor writes its C rows (its box once per class, like the reference's tile over classes).
__global__ __launch_bounds__(kHistThreads) void raw_detections_kernel(const DetArgs d, int n_items, float* ids,
                                                                      float* scores, float* bboxes,
                                                                      int32_t* keep_idx) {
  const int b = blockIdx.y;
  const int it = blockIdx.x * kHistThreads + threadIdx.x;
  Item im;
  if (it >= n_items || !locate(d, b, it, im)) return;
  const float conf = vy_sigmoidf(im.p[4]);
  float x1, y1, x2, y2;
  decode_box(d, im, x1, y1, x2, y2);
  for (int c = 0; c < d.C; ++c) {
    const size_t o = (size_t)b * d.n_cand + (size_t)(im.cand0 + c * im.cstride);
    ids[o] = (float)c;
    scores[o] = vy_sigmoidf(im.p[5 + c]) * conf;
    bboxes[o * 4 + 0] = x1;
    bboxes[o * 4 + 1] = y1;
    bboxes[o * 4 + 2] = x2;
    bboxes[o * 4 + 3] = y2;
    if (keep_idx) keep_idx[o] = im.cand0 + c * im.cstride;
  }
}

}  // namespace

hipError_t vy_launch_raw_detections(const DetArgs& a, float* ids, float* scores, float* bboxes, int32_t* keep_idx,
                                    hipStream_t s) {
  int n_items = 0;
  for (int i = 0; i < 3; ++i) n_items += a.head[i].H * a.head[i].W * 3;
  dim3 grid((n_items + kHistThreads - 1) / kHistThreads, a.B);
  hipLaunchKernelGGL(raw_detections_kernel, grid, dim3(kHistThreads), 0, s, a, n_items, ids, scores, bboxes, keep_idx);
  return hipGetLastError();
}

size_t vy_det_scratch_bytes(int B, int n_items, int C) {
  return align256(sizeof(SelState) * (size_t)B) + align256(sizeof(uint32_t) * (size_t)B * kBins) +
         align256(sizeof(Entry) * (size_t)B * VY_NMS_MAX_TOPK) +
         align256(sizeof(unsigned long long) * (size_t)B * kListCap) + align256(sizeof(float) * (size_t)B * C * n_items);
}

hipError_t vy_launch_detect(const DetArgs& a, void* scratch, float* ids, float* scores, float* bboxes,
                            int32_t* keep_idx, hipStream_t s) {
  if (a.n_cand >= (1 << kIdxBits)) return hipErrorInvalidValue;
  // nms_topk <= 0 or > VY_NMS_MAX_TOPK: the chunked kernel (kept rows in LDS for outputs of <= VY_NMS_MAX_TOPK rows,
  // read back from the output itself for longer ones)
  const bool chunked = a.topk <= 0 || a.topk > VY_NMS_MAX_TOPK;
  // rows of the output: post_nms, else nms_topk (no slice, yolo3.py:1201-1202), else — both "disabled" — all N*C rows
  const int rows = a.post_nms > 0 ? a.post_nms : (a.topk > 0 ? a.topk : a.n_cand);
  // state + histogram region back to zero (entries need no clearing)
  hipError_t e = hipMemsetAsync(scratch, 0,
                                align256(sizeof(SelState) * (size_t)a.B) +
                                    align256(sizeof(uint32_t) * (size_t)a.B * kBins),
                                s);
  if (e != hipSuccess) return e;
  int n_items = 0;
  for (int i = 0; i < 3; ++i) n_items += a.head[i].H * a.head[i].W * 3;
  const int per_block = kHistThreads * kItemsPerThread;
  dim3 grid((n_items + per_block - 1) / per_block, a.B);
  int hblocks = 0;
  for (int i = 0; i < 3; ++i) {
    hblocks += hist_blocks_of(a.head[i].H * a.head[i].W, a.head[i].cs);
    if (a.head[i].cs > kTileFloats || (a.head[i].cs & 3) || (a.head[i].co & 3)) return hipErrorInvalidValue;
  }
  const dim3 hgrid(hblocks, a.B);
  if (chunked) {  // every valid candidate (or the topk > 1024 best) goes through NMS: pass 0 only fills the score cache
    hipLaunchKernelGGL(hist_kernel, hgrid, dim3(kHistThreads), 0, s, a, scratch, 0, n_items);
    hipLaunchKernelGGL(nms_all_kernel, dim3(a.B), dim3(kAllThreads), 0, s, a, scratch, n_items, rows,
                       a.topk > 0 ? a.topk : 0, ids, scores, bboxes, keep_idx);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(hist_kernel, hgrid, dim3(kHistThreads), 0, s, a, scratch, 0, n_items);
  hipLaunchKernelGGL(select_kernel, dim3(a.B), dim3(kBins), 0, s, a, scratch, 0);
  const int c_groups = (a.C + 7) / 8;
  if ((long long)grid.x * a.B < 256) {
    hipLaunchKernelGGL(compact_kernel<1>, dim3((n_items + kHistThreads - 1) / kHistThreads, a.B, c_groups), dim3(kHistThreads), 0, s,
                       a, scratch, n_items, 1);
  } else {
    hipLaunchKernelGGL(compact_kernel<kItemsPerThread>, grid, dim3(kHistThreads), 0, s, a, scratch, n_items, c_groups);
  }
  hipLaunchKernelGGL(refine_kernel, dim3(a.B), dim3(kBins), 0, s, a, scratch, n_items);
  hipLaunchKernelGGL(sort_nms_kernel, dim3(a.B), dim3(kNmsThreads), 0, s, a, scratch, rows, ids, scores, bboxes,
                     keep_idx);
  return hipGetLastError();
}
