// Host-side check of videoyolo_amd/csrc/conv_cost_model.h (g++, CPU test suite): the (block tile, stream-K?) decisions for
// the conv shapes of BASELINE.json's workloads, as measured and adopted on the MI355X in round 3, and the model's
// structural properties.  A change of a constant that moves one of these decisions has to be re-measured, not slipped in.
// Output: one line per shape "M N K -> BMxBN[sk]"; exit code 1 on a violated property.
#include <cstdio>

#include "../videoyolo_amd/csrc/conv_cost_model.h"

int main() {
  const VySkPolicy on = {true, 0.03, 7.5}, off = {false, 0.03, 7.5};
  struct Shape {
    long long M;
    int N;
    double K;
  };
  // 608x608 batch 64 (configs[1]), 416x416 batch 16 forward (configs[2]), 608x608 batch 1
  const Shape shapes[] = {
      {64ll * 304 * 304, 64, 288},  {64ll * 304 * 304, 32, 64},   {64ll * 152 * 152, 128, 576}, {64ll * 152 * 152, 64, 128},
      {64ll * 76 * 76, 256, 1152},  {64ll * 76 * 76, 128, 256},   {64ll * 38 * 38, 512, 2304},  {64ll * 38 * 38, 256, 512},
      {64ll * 19 * 19, 1024, 4608}, {64ll * 19 * 19, 512, 1024},  {64ll * 19 * 19, 75, 1024},   {16ll * 52 * 52, 256, 1152},
      {16ll * 26 * 26, 512, 2304},  {16ll * 13 * 13, 1024, 4608}, {16ll * 13 * 13, 512, 1024},  {16ll * 52 * 52, 128, 256},
      {1ll * 76 * 76, 256, 1152},   {1ll * 19 * 19, 1024, 4608},
  };
  for (const Shape& s : shapes) {
    int bm = 0, bn = 0, bm0 = 0, bn0 = 0;
    bool sk = false, sk0 = true;
    const double t = vy_select_tile(s.M, s.N, s.K, on, &bm, &bn, &sk);
    const double t0 = vy_select_tile(s.M, s.N, s.K, off, &bm0, &bn0, &sk0);
    printf("%lld %d %.0f -> %dx%d%s", s.M, s.N, s.K, bm, bn, sk ? "sk" : "");
    // conv mode VY_CONV_SPLIT_BF16X3 (round 4): which kernel the launch goes to (vy_conv_split_pays: the split model's
    // time against 0.97 x the exact model's), and the split kernel's tile / k-split
    if (s.N % 64 == 0) {
      int sbm = 0, sbn = 0, sks = 0;
      const double ts = vy_predict_split(s.M, s.N, s.K, 64, &sbm, &sbn, &sks);
      if (ts < 0.97 * t) printf("   | split %dx%d%s%.0d", sbm, sbn, sks > 1 ? " k" : "", sks > 1 ? sks : 0);
      else printf("   | exact");
      if (sks < 1 || (sks > 1 && (s.K / 16) / sks < 6)) return fprintf(stderr, "k-split with slices shorter than 6 k-steps\n"), 1;
      int b1, b2, k1;
      if (vy_predict_split(s.M, s.N, s.K, 1, &b1, &b2, &k1) < ts - 1e-9 || k1 != 1)
        return fprintf(stderr, "allowing k-split made the split prediction worse\n"), 1;
    }
    // ... and whether its 3x3 stride-1 cell goes to the Winograd F(2, 3) instance instead (vy_conv_wino_pays: its model
    // against 0.97 x the better of the other two; pairs = M / 2: even widths)
    if (s.N % 128 == 0 && (long long)s.K % 288 == 0) {
      int sbm = 0, sbn = 0, sks = 0;
      const double ts = vy_predict_split(s.M, s.N, s.K, 64, &sbm, &sbn, &sks);
      const double tw = vy_predict_wino(s.M / 2, s.N, (int)(s.K / 9));
      printf(tw < 0.97 * (ts < t ? ts : t) ? "   | wino" : "   | -");
    }
    printf("\n");
    if (sk0) return fprintf(stderr, "stream-K chosen although not allowed\n"), 1;
    if (t > t0 * (1.0 + 1e-12)) return fprintf(stderr, "allowing stream-K made the prediction worse\n"), 1;
    if (s.N <= 32 && !(bm == 128 && bn == 32 && !sk)) return fprintf(stderr, "N <= 32 must use the 128x32 tile\n"), 1;
    if (s.N > 32 && !vy_tile_model(bm, bn)) return fprintf(stderr, "tile without a model\n"), 1;
    if (sk) {  // a stream-K launch has more tiles than blocks, and its predicted saving clears the threshold
      const VyTileModel* m = vy_tile_model(bm, bn);
      const long long tiles = ((s.M + bm - 1) / bm) * ((s.N + bn - 1) / bn);
      const long long per_cu = tiles / 256 < m->resident ? tiles / 256 : m->resident;
      bool dummy;
      const double plain = vy_predict_launch(s.M, s.N, s.K, *m, off, &dummy);
      if (per_cu < 1 || tiles <= 256 * per_cu) return fprintf(stderr, "stream-K launch without a cut tile\n"), 1;
      if (plain - t < 0.03 * plain - 1e-9) return fprintf(stderr, "stream-K below its threshold\n"), 1;
    }
  }
  // ---- another chip: the models count their rounds in the device's CU count, and every choice
  // BETWEEN kernels is refused where the constants were not fitted.  128 CUs = the MI355X in DPX mode.
  if (!vy_model_fitted(256) || vy_model_fitted(128) || vy_model_fitted(64) || vy_model_fitted(304))
    return fprintf(stderr, "vy_model_fitted must hold for 256 CUs only\n"), 1;
  printf("-- 128 CUs\n");
  for (const Shape& s : shapes) {
    int bm = 0, bn = 0, b2 = 0, n2 = 0;
    bool sk = false, sk2 = false;
    const double t128 = vy_select_tile(s.M, s.N, s.K, off, &bm, &bn, &sk, 128);
    const double t256 = vy_select_tile(s.M, s.N, s.K, off, &b2, &n2, &sk2, 256);
    printf("%lld %d %.0f -> %dx%d\n", s.M, s.N, s.K, bm, bn);
    if (s.N > 32 && t128 < t256 * (1.0 - 1e-12)) return fprintf(stderr, "half the CUs predicted faster\n"), 1;
    // a launch that fills 256 CUs for many rounds takes twice as long on 128
    const long long tiles = ((s.M + bm - 1) / bm) * ((s.N + bn - 1) / bn);
    if (s.N > 32 && tiles >= 64 * 256 && bm == b2 && bn == n2 && (t128 < 1.9 * t256 || t128 > 2.1 * t256))
      return fprintf(stderr, "a long launch must scale with the CU count (%.1f vs %.1f)\n", t128, t256), 1;
    if (s.N % 64 == 0) {
      int sbm, sbn, sks;
      const double a = vy_predict_split(s.M, s.N, s.K, 64, &sbm, &sbn, &sks, 128), b = vy_predict_split(s.M, s.N, s.K, 64, &sbm, &sbn, &sks, 256);
      if (a < b * (1.0 - 1e-12)) return fprintf(stderr, "split model: half the CUs predicted faster\n"), 1;
    }
    if (s.N % 128 == 0 && (long long)s.K % 288 == 0 &&
        vy_predict_wino(s.M / 2, s.N, (int)(s.K / 9), 128) < vy_predict_wino(s.M / 2, s.N, (int)(s.K / 9), 256) * (1.0 - 1e-12))
      return fprintf(stderr, "Winograd model: half the CUs predicted faster\n"), 1;
  }
  return 0;
}
