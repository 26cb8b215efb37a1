// conv_cost_model.h — the launch cost model of the implicit-GEMM conv: which block tile a launch uses and whether it runs
// as a chain-preserving stream-K launch.  Host-only, plain C++ (no HIP types): conv_igemm.hip includes it, and so does
// tests/conv_cost_model_check.cpp, which the CPU test suite compiles with g++ to pin the decisions for BASELINE's shapes.
//
// Measured on the MI355X (tools/train_layers.sh with VY_CONV_FORCE, 416x416 batch 16 and 608x608 batch 64): a launch of T
// tiles takes ceil(T / 256) x (alpha(tile) x K + O(tile)) — every CU works through its share of the tiles at a rate that
// does not depend on how many blocks it holds (2 resident 128x128 blocks, 3 of 128x64, 4 of 64x64), and what is lost is
// the last, partly filled round of the 256 CUs.  Fitted per tile, in microseconds (refitted after the k-loop lost its
// vector instructions): alpha = 0.0543 / 0.0273 / 0.0145 per unit of K (the 128x64 tile costs exactly half of 128x128),
// fixed part O = 4.5 / 2.6 / 1.4 (prologue tables + the epilogue of 64 / 32 / 16 accumulator registers per lane):
// short-K 1x1 layers prefer the small tiles, long-K 3x3 layers the large one.  A stream-K launch (hybrid schedule,
// sk_schedule.h) takes T / 256 rounds, not rounded up, plus one hand-off: 7.5 us fitted on the short-K launches, which
// lose exactly that.  (A two-wave 64x32 tile was tried for the 13x13 maps at batch 16, which are short of blocks: 461 vs
// 337 us on the K = 9216 data gradients.)
//
// The chip.  Every constant below was fitted on the MI355X in SPX mode: 256 CUs.  The models take the CU count of the
// device the launch goes to (`cus`: vy_cu_count(), hipDeviceAttributeMultiprocessorCount) and count their rounds in it —
// a round is what ONE CU does with its share, which does not depend on how many CUs there are.  What does depend on the
// chip is every comparison BETWEEN kernels (the split and Winograd instances were fitted against the exact kernel under
// this part's power cap and L2 / LDS-DMA rates): vy_model_fitted(cus) is false for any other count, and the callers then
// stay on the exact kernel's own tile model (vy_conv_split_pays / vy_conv_wino_pays refuse, as stream-K does through
// vy_sk_verify_topology) — a partition mode (DPX 128, QPX 64, CPX 32 CUs) or a CU-masked queue gets correct, unsurprising
// launches instead of choices that silently refer to a different chip.
#pragma once

#include <algorithm>

#define VY_MODEL_CUS 256
inline bool vy_model_fitted(int cus) { return cus == VY_MODEL_CUS; }

struct VyTileModel {
  int bm, bn;
  double alpha, fixed;
  int resident;  // blocks per CU
};
static const VyTileModel kVyTileModels[3] = {{128, 128, 0.0543, 4.5, 2}, {128, 64, 0.0273, 2.6, 3}, {64, 64, 0.0145, 1.4, 4}};

inline const VyTileModel* vy_tile_model(int bm, int bn) {
  for (const VyTileModel& t : kVyTileModels)
    if (t.bm == bm && t.bn == bn) return &t;
  return nullptr;
}

struct VySkPolicy {
  bool allowed;     // stream-K instances may be used for this launch at all
  double min_gain;  // fraction of the plain launch the predicted saving has to reach (0.03)
  double cost;      // microseconds of one hand-off (7.5)
};

// predicted time (microseconds) of a launch of M x N outputs with reduction length K on tile t; *use_sk: as stream-K
inline double vy_predict_launch(long long M, int N, double K, const VyTileModel& t, const VySkPolicy& p, bool* use_sk,
                                int cus = VY_MODEL_CUS) {
  const long long tiles = ((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn);
  const double t_round = t.alpha * K + t.fixed;
  // a block that has a CU to itself (one wave per SIMD) runs over the model: nobody covers its LDS-DMA latency.  Small
  // tiles 10-19 % (their four-stage instance); a 128x128 block 25-33 % (k-loop of 20.6 us for 15.5 in a pair,
  // tools/probe/conv_tile_trace.hip; the 26x26 training layers: 195 us as 128x128 stream-K with one block per CU against
  // 182 us as 128x64 stream-K with two)
  const double lone = t.bm * t.bn < 128 * 128 ? 1.12 : 1.25;
  double t_plain = (double)((tiles + cus - 1) / cus) * t_round;
  if (tiles < cus) t_plain *= lone;
  *use_sk = false;
  if (p.allowed && tiles > cus) {
    const long long per_cu = std::min<long long>(t.resident, tiles / cus);  // a share is at least one tile
    if (tiles > cus * per_cu) {
      double t_sk = (double)tiles / (double)cus * t_round;
      if (per_cu == 1) t_sk *= lone;
      t_sk += p.cost;
      if (t_plain - t_sk >= p.min_gain * t_plain) {
        *use_sk = true;
        return t_sk;
      }
    }
  }
  return t_plain;
}

// the block tile with the smallest predicted time (a smaller tile has to be better by 0.5 %); N <= 32 has its own tile
inline double vy_select_tile(long long M, int N, double K, const VySkPolicy& p, int* bm, int* bn, bool* use_sk,
                             int cus = VY_MODEL_CUS) {
  if (N <= 32) {
    *bm = 128, *bn = 32, *use_sk = false;
    return 0.0;
  }
  double best = 1e300;
  for (const VyTileModel& c : kVyTileModels) {
    if (c.bn == 128 && N <= 64) continue;
    bool sk;
    const double t = vy_predict_launch(M, N, K, c, p, &sk, cus);  // as a plain or a stream-K launch, whichever will be used
    if (t < best * 0.995) {
      best = t;
      *bm = c.bm;
      *bn = c.bn;
      *use_sk = sk;
    }
  }
  return best;
}

// ---- the split-fp32 instance (conv_split.hip; conv mode VY_CONV_SPLIT_BF16X3) ------------------------------------------
// Same form, fitted on tools/probe/run_split_batch_sweep.sh (batch 1 ... 64 on six layer shapes): a CU works through its
// tiles at alpha x K + O each whether it holds one block or two (two 128x128 blocks take 1.19 us per 16-channel k-step
// together, a lone one 0.66 us), so a launch takes ceil(tiles / 256) rounds; a block alone on its CU runs 10 % (128-row
// tiles of 128 channels) to 40 % (128 x 64) over the model.  alpha per unit of K: 0.0372 (128x128; the exact kernel: 0.0543),
// 0.0227 (128x64), 0.043 (256x64, the tile of the 64-channel layers).  With few tiles the exact kernel wins — it has the
// 64x64 tile, four LDS stages and stream-K — e.g. every layer of a single 608x608 frame but the 152x152 / 304x304 ones.
struct VySplitModel {
  int bm, bn;
  double alpha, fixed, lone;
};
static const VySplitModel kVySplitModels[3] = {{128, 128, 0.0372, 3.0, 1.10}, {128, 64, 0.0227, 2.0, 1.40}, {256, 64, 0.043, 4.5, 1.10}};

// predicted time (microseconds) of the launch on the split-fp32 kernel, the tile and the k-split it would use; N % 64 == 0.
// k-split S > 1 (at most max_ksplit: what the slab scratch holds): tiles x S blocks of K / S each plus the finish
// launch — for launches that leave most CUs without a block (a single frame's 19x19 layer: 24 tiles of 288 k-steps).
// Fitted on tools/probe/run_split_ksplit_sweep.sh: the second launch, the slab round trip and the short k-loops'
// pipeline fill cost about 9 us (19x19, K = 4608, one frame: 190 us unsplit, 36.7 us as 8 slices; the exact kernel 75 us);
// every launch carries 4 us of fill / drain (a single frame's 152x152 layers: 38.6 us on 181 lone blocks, the exact kernel 34.3).
inline double vy_predict_split(long long M, int N, double K, int max_ksplit, int* bm, int* bn, int* ksplit, int cus = VY_MODEL_CUS) {
  double best = 1e300;
  *ksplit = 1;
  for (const VySplitModel& c : kVySplitModels) {
    if (N % c.bn != 0) continue;
    if (c.bm == 256 && N % 128 == 0) continue;  // the 256-row tile is the 64-channel layers' only
    const long long tiles = ((M + c.bm - 1) / c.bm) * (N / c.bn);
    const int steps = (int)(K / 16.0);
    for (int S = 1; S <= max_ksplit && S <= 32; ++S) {
      if (S > 1 && (tiles * (S - 1) >= 2 * cus || steps / S < 6)) break;  // only while CUs have room and slices stay long enough
      const long long blocks = tiles * S;
      double t = (double)((blocks + cus - 1) / cus) * (c.alpha * K / S + c.fixed);
      if (blocks <= cus) t *= c.lone;
      t += 4.0;               // per launch: pipeline fill and drain of a kernel that prefetches two k-steps ahead
      if (S > 1) t += 5.0;    // the finish launch and the slab round trip
      if (t < best * 0.995) {
        best = t;
        *bm = c.bm;
        *bn = c.bn;
        *ksplit = S;
      }
    }
  }
  return best;
}

// ---- the Winograd F(2, 3) instance of the split conv (conv_wino.hip: 64 pixel pairs x 128 channels per block, two blocks per
// CU, no k-split) -----------------------------------------------------------------------------------------------------------
// Fitted on tools/layer_profile.py with VY_SPLIT_WINO=2 / 0 at 608x608 and 416x416, batch 1 ... 64: a block alone on its CU
// takes 0.40 us per input channel + 10 (launch included); two resident blocks 0.62 us per channel + 7 together, a launch of
// more than 256 blocks that many rounds of 512 (whole rounds up to four, then the fraction: the tail of a long launch
// overlaps).  Kc = input channels (the four GEMMs' K is 3 Kc each).
inline double vy_predict_wino(long long pairs, int N, int Kc, int cus = VY_MODEL_CUS) {
  const long long blocks = ((pairs + 63) / 64) * (N / 128), pair_round = 2ll * cus;  // two resident blocks per CU
  if (blocks <= cus) return 0.40 * Kc + 10.0;
  if (blocks <= 2 * pair_round) return (double)((blocks + pair_round - 1) / pair_round) * (0.55 * Kc + 6.0) + 4.0;  // (short launches: the chip is not yet held by its power cap)
  const double rounds = blocks <= 4 * pair_round ? (double)((blocks + pair_round - 1) / pair_round) : (double)blocks / (double)pair_round;
  return rounds * (0.62 * Kc + 7.0) + 4.0;
}
