// Where does a conv tile's time go, and what limits the matrix pipe?  Runs ONE synthetic conv launch through the
// product kernel (compiled here with -DVY_CONV_TRACE: thread 0 of every block stamps s_memrealtime after the
// prologue, the k-loop and the epilogue) on all-zero or random-normal operands and prints the launch time, the
// phases' statistics and, per CU, how long no / one / two blocks were inside a k-loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVY_CONV_TRACE -I include -o conv_tile_trace \
//       tools/probe/conv_tile_trace.hip && ./conv_tile_trace B H Cin Cout k [stride=1] [res=0] [random_data=0]
// Round-2 finding (gpurun_out/NOTES_r2.txt): a block that has a CU's matrix pipe to itself runs its
// k-loop at 97 % of the fp32 MFMA peak on zero operands and 7 % slower on random ones, DMA and LDS reads included or
// not — the launch is limited by the chip's power management, not by instruction issue.
#include "../../videoyolo_amd/csrc/conv_igemm.hip"

#include <algorithm>
#include <cstring>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static hipError_t fill_normal(float* d, size_t n, float scale) {
  std::vector<float> h(n);
  unsigned long long st = 88172645463325252ull;
  for (size_t i = 0; i < n; ++i) {
    float acc = 0.f;
    for (int j = 0; j < 4; ++j) {  // sum of 4 uniforms: close enough to normal for a power experiment
      st ^= st << 13;
      st ^= st >> 7;
      st ^= st << 17;
      acc += (float)(st >> 40) * (1.0f / 16777216.0f) - 0.5f;
    }
    h[i] = acc * 1.7320508f * scale;
  }
  return hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s B H Cin Cout k [stride=1] [res=0] [random_data=0]\n", argv[0]);
    return 2;
  }
  const int B = atoi(argv[1]), H = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]), k = atoi(argv[5]);
  const int stride = argc > 6 ? atoi(argv[6]) : 1, res = argc > 7 ? atoi(argv[7]) : 0;
  const int random_data = argc > 8 ? atoi(argv[8]) : 0;
  const int Ho = H / stride;
  const size_t in_n = (size_t)B * (H + 2) * (H + 2) * Cin, out_n = (size_t)B * (Ho + 2) * (Ho + 2) * Cout;
  const size_t w_n = (size_t)Cout * k * k * Cin;
  float *in, *out, *w, *sc, *sh, *rs = nullptr;
  CK(hipMalloc(&in, in_n * 4));
  CK(hipMalloc(&out, out_n * 4));
  CK(hipMalloc(&w, w_n * 4));
  CK(hipMalloc(&sc, Cout * 4));
  CK(hipMalloc(&sh, Cout * 4));
  CK(hipMemset(in, 0, in_n * 4));
  CK(hipMemset(w, 0, w_n * 4));
  CK(hipMemset(sc, 0, Cout * 4));
  CK(hipMemset(sh, 0, Cout * 4));
  if (res) {
    CK(hipMalloc(&rs, out_n * 4));
    CK(hipMemset(rs, 0, out_n * 4));
  }
  if (random_data) {  // (borders included: this is a timing experiment)
    CK(fill_normal(in, in_n, 1.0f));
    CK(fill_normal(w, w_n, 0.05f));
    CK(fill_normal(sc, Cout, 1.0f));
  }
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.in = in; a.w = w; a.scale = sc; a.shift = sh; a.res = rs; a.out = out;
  a.B = B; a.LH = Ho; a.LW = Ho; a.M = B * Ho * Ho;
  a.a_Hp = H + 2; a.a_Wp = H + 2; a.a_cs = Cin; a.a_co = 0; a.a_s = stride; a.a_oy = a.a_ox = 1;
  a.Kc = Cin; a.ntaps = k * k;
  for (int t = 0; t < a.ntaps; ++t) {
    a.tap_dy[t] = (signed char)(k == 3 ? t / 3 - 1 : 0);
    a.tap_dx[t] = (signed char)(k == 3 ? t % 3 - 1 : 0);
    a.tap_w[t] = (unsigned char)t;
  }
  a.w_taps = k * k; a.w_cin = Cin; a.w_cout = Cout; a.N = Cout;
  a.o_Hp = Ho + 2; a.o_Wp = Ho + 2; a.o_cs = Cout; a.o_co = 0; a.o_s = 1; a.o_oy = a.o_ox = 1; a.ups = 1;
  a.r_cs = Cout; a.r_co = 0; a.leaky = 1; a.dgrad = 0;
  {  // stream-K scratch, as a net's workspace provides it (zeroed: all flags down)
    void* skp;
    CK(hipMalloc(&skp, (size_t)VY_SK_PARTIAL_BYTES + (size_t)VY_SK_FLAGS * 4));
    CK(hipMemset(skp, 0, (size_t)VY_SK_PARTIAL_BYTES + (size_t)VY_SK_FLAGS * 4));
    a.sk_flags = (unsigned*)skp;
    a.sk_partials = (float*)((char*)skp + (size_t)VY_SK_FLAGS * 4);
    a.sk_bytes = VY_SK_PARTIAL_BYTES;
    a.sk_nflags = VY_SK_FLAGS;
  }
  int bm, bn;
  vy_conv_cfg(a, &bm, &bn);
  const bool streamk = vy_conv_streamk(a);
  const long long tiles = (long long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn);
  unsigned long long* trace;
  CK(hipMalloc(&trace, (size_t)tiles * 8 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  a.trace = nullptr;
  for (int i = 0; i < 3; ++i) CK(vy_launch_conv_igemm(a, 0));
  CK(hipEventRecord(e0, 0));
  const int reps = getenv("VY_TRACE_REPS") ? atoi(getenv("VY_TRACE_REPS")) : 20;  // (clock_under_load.sh: seconds of the same launch)
  for (int i = 0; i < reps; ++i) CK(vy_launch_conv_igemm(a, 0));
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps, K = (double)k * k * Cin;
  const double gflop = 2.0 * a.M * (double)Cout * K * 1e-9;
  printf("conv B=%d H=%d Cin=%d Cout=%d k=%d s=%d res=%d %s | M=%d N=%d K=%.0f tile %dx%d%s tiles=%lld | %.1f us  %.1f TF\n",
         B, H, Cin, Cout, k, stride, res, random_data ? "random" : "zeros ", a.M, a.N, K, bm, bn, streamk ? "sk" : "", tiles, us,
         gflop / us * 1e3);
  if (streamk) return 0;  // (the per-block trace below assumes one tile per block)
  CK(hipMemset(trace, 0, (size_t)tiles * 8 * 8));
  a.trace = trace;
  CK(vy_launch_conv_igemm(a, 0));
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> tr((size_t)tiles * 8);
  CK(hipMemcpy(tr.data(), trace, tr.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> pro, kloop, epi;
  unsigned long long tmin = ~0ull, tmax = 0;
  struct Ev {
    unsigned long long t;
    int d;
  };
  std::vector<std::vector<Ev>> cu(8 * 256);
  std::vector<int> ntile(8 * 256, 0);
  for (long long b = 0; b < tiles; ++b) {
    const unsigned long long* r = &tr[(size_t)b * 8];
    if (!r[3]) continue;
    pro.push_back((r[1] - r[0]) * 0.01);
    kloop.push_back((r[2] - r[1]) * 0.01);
    epi.push_back((r[3] - r[2]) * 0.01);
    tmin = std::min(tmin, r[0]);
    tmax = std::max(tmax, r[3]);
    const int key = (int)((r[5] & 7) * 256 + ((r[4] >> 8) & 0xff));
    cu[key].push_back({r[1], +1});
    cu[key].push_back({r[2], -1});
    ++ntile[key];
  }
  double occ[4] = {0, 0, 0, 0}, span = 0;
  int ncu = 0, lo = 1 << 30, hi = 0;
  for (size_t c = 0; c < cu.size(); ++c) {
    if (cu[c].empty()) continue;
    ++ncu;
    lo = std::min(lo, ntile[c]);
    hi = std::max(hi, ntile[c]);
    std::sort(cu[c].begin(), cu[c].end(), [](const Ev& x, const Ev& y) { return x.t < y.t || (x.t == y.t && x.d < y.d); });
    int lvl = 0;
    unsigned long long prev = tmin;
    for (const Ev& e : cu[c]) {
      occ[lvl < 3 ? lvl : 3] += (e.t - prev) * 0.01;
      prev = e.t;
      lvl += e.d;
    }
    occ[0] += (tmax - prev) * 0.01;
    span += (tmax - tmin) * 0.01;
  }
  printf("  traced launch %.1f us on %d CUs, %d..%d tiles per CU; a tile's MFMA work at the fp32 peak: %.2f us\n",
         (tmax - tmin) * 0.01, ncu, lo, hi, 2.0 * bm * bn * K / 614.4e3);
  printf("  CU time with 0 / 1 / 2 / 3+ blocks inside a k-loop: %.1f%% / %.1f%% / %.1f%% / %.1f%%\n", 100 * occ[0] / span,
         100 * occ[1] / span, 100 * occ[2] / span, 100 * occ[3] / span);
  auto stat = [](const char* name, std::vector<double>& v) {
    if (v.empty()) return;
    std::sort(v.begin(), v.end());
    double s = 0;
    for (double x : v) s += x;
    printf("  %-8s n=%6zu mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  max %7.2f us\n", name, v.size(), s / v.size(),
           v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
  };
  // where the dispatcher puts the first blocks of XCD 0 (blockIdx 0, 8, 16, ...): CU id and start time
  printf("  first blocks of XCD 0: (block, cu, start us)");
  for (long long b = 0; b < tiles && b < 8 * 80; b += 8) {
    const unsigned long long* r = &tr[(size_t)b * 8];
    if (!r[3]) continue;
    printf(" (%lld,%d,%.1f)", b, (int)((r[4] >> 8) & 0xff), (r[0] - tmin) * 0.01);
  }
  printf("\n");
  stat("prologue", pro);
  stat("k-loop", kloop);
  stat("epilogue", epi);
  return 0;
}
