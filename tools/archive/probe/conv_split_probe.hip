// Split-fp32 (bf16 x 3) conv against the exact fp32 conv on one layer shape: speed and distance.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-inline-asm -I include -o conv_split_probe \
//       tools/probe/conv_split_probe.hip && ./conv_split_probe B H Cin Cout k [stride=1] [res=0] [reps=20]
// Prints, for random-normal operands: launch time and TFLOP/s (fp32-equivalent) of both kernels, the largest
// difference between them, and both kernels' error against a float64 evaluation of 4096 sampled outputs.
#include "../../videoyolo_amd/csrc/conv_igemm.hip"
#include "../../videoyolo_amd/csrc/conv_split.hip"

#include <cmath>
#include <cstring>
#include <vector>

#define CK(x)                                                 \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if (e_ != hipSuccess) {                                   \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      return 1;                                               \
    }                                                         \
  } while (0)

static void fill_normal(std::vector<float>& h, float scale, unsigned long long seed) {
  unsigned long long st = seed;
  for (size_t i = 0; i < h.size(); ++i) {
    float acc = 0.f;
    for (int j = 0; j < 4; ++j) {
      st ^= st << 13;
      st ^= st >> 7;
      st ^= st << 17;
      acc += (float)(st >> 40) * (1.0f / 16777216.0f) - 0.5f;
    }
    h[i] = acc * 1.7320508f * scale;
  }
}

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s B H Cin Cout k [stride=1] [res=0] [reps=20]\n", argv[0]);
    return 2;
  }
  const int B = atoi(argv[1]), H = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]), k = atoi(argv[5]);
  const int stride = argc > 6 ? atoi(argv[6]) : 1, res = argc > 7 ? atoi(argv[7]) : 0;
  const int reps = argc > 8 ? atoi(argv[8]) : 20;
  const int dgrad = getenv("VY_PROBE_DGRAD") ? atoi(getenv("VY_PROBE_DGRAD")) : 0;  // 1: the data gradient of that conv (stride 1)
  const int Ho = (H + stride - 1) / stride;
  const size_t in_n = (size_t)B * (H + 2) * (H + 2) * Cin, out_n = (size_t)B * (Ho + 2) * (Ho + 2) * Cout;
  const size_t w_n = (size_t)Cout * k * k * Cin;
  std::vector<float> h_in(in_n), h_w(w_n), h_sc(Cout), h_sh(Cout), h_rs(res ? out_n : 0);
  fill_normal(h_in, 1.0f, 88172645463325252ull);
  fill_normal(h_w, 1.0f / sqrtf((float)(k * k * Cin)), 1234567ull);
  fill_normal(h_sc, 0.2f, 99ull);
  for (auto& v : h_sc) v += 1.0f;
  fill_normal(h_sh, 0.5f, 777ull);
  if (res) fill_normal(h_rs, 1.0f, 4242ull);
  if (getenv("VY_PROBE_ZERO")) {  // power experiment: all-zero operands (same instruction stream, no toggling)
    std::fill(h_in.begin(), h_in.end(), 0.f);
    std::fill(h_w.begin(), h_w.end(), 0.f);
  }
  // zero borders, as the planes have them
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H + 2; ++y)
      for (int x = 0; x < H + 2; ++x)
        if (y == 0 || x == 0 || y == H + 1 || x == H + 1)
          memset(&h_in[(((size_t)b * (H + 2) + y) * (H + 2) + x) * Cin], 0, Cin * 4);
  float *in, *out, *out2, *w, *sc, *sh, *rs = nullptr;
  void* wimg;
  CK(hipMalloc(&in, in_n * 4));
  CK(hipMalloc(&out, out_n * 4));
  CK(hipMalloc(&out2, out_n * 4));
  CK(hipMalloc(&w, w_n * 4));
  CK(hipMalloc(&sc, Cout * 4));
  CK(hipMalloc(&sh, Cout * 4));
  CK(hipMalloc(&wimg, vy_split_weight_bytes(Cout, k * k, Cin)));
  CK(hipMemcpy(in, h_in.data(), in_n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h_w.data(), w_n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, h_sc.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, h_sh.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemset(out, 0, out_n * 4));
  CK(hipMemset(out2, 0, out_n * 4));
  if (res) {
    CK(hipMalloc(&rs, out_n * 4));
    CK(hipMemcpy(rs, h_rs.data(), out_n * 4, hipMemcpyHostToDevice));
  }
  CK(vy_launch_split_weights(w, wimg, Cout, k * k, Cin, 0));
  void* wimg_d = nullptr;
  if (dgrad) {
    if (stride != 1 || H != Ho) return fprintf(stderr, "dgrad probe: stride 1 only\n"), 2;
    CK(hipMalloc(&wimg_d, vy_split_weight_dgrad_bytes(Cout, k * k, Cin)));
    CK(vy_launch_split_weights_dgrad(w, wimg_d, Cout, k * k, Cin, 0));
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
  a.w_split = wimg;
  {
    void* skp;
    CK(hipMalloc(&skp, (size_t)VY_SK_PARTIAL_BYTES + (size_t)VY_SK_FLAGS * 4));
    CK(hipMemset(skp, 0, (size_t)VY_SK_PARTIAL_BYTES + (size_t)VY_SK_FLAGS * 4));
    a.sk_flags = (unsigned*)skp;
    a.sk_partials = (float*)((char*)skp + (size_t)VY_SK_FLAGS * 4);
    a.sk_bytes = VY_SK_PARTIAL_BYTES;
    a.sk_nflags = VY_SK_FLAGS;
    void* slabs;
    CK(hipMalloc(&slabs, (size_t)VY_SK_PARTIAL_BYTES));
    a.splitk_slabs = (float*)slabs;
    a.splitk_bytes = VY_SK_PARTIAL_BYTES;
  }
  float *dz = nullptr, *gin = nullptr, *gin2 = nullptr;
  if (dgrad) {  // dz = the forward probe's input tensor reinterpreted is no use (channel counts differ): own planes
    const size_t dz_n = (size_t)B * (H + 2) * (H + 2) * Cout, g_n = (size_t)B * (H + 2) * (H + 2) * Cin;
    std::vector<float> h_dz(dz_n);
    fill_normal(h_dz, 1.0f, 31337ull);
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < H + 2; ++y)
        for (int x = 0; x < H + 2; ++x)
          if (y == 0 || x == 0 || y == H + 1 || x == H + 1) memset(&h_dz[(((size_t)b * (H + 2) + y) * (H + 2) + x) * Cout], 0, Cout * 4);
    CK(hipMalloc(&dz, dz_n * 4));
    CK(hipMalloc(&gin, g_n * 4));
    CK(hipMalloc(&gin2, g_n * 4));
    CK(hipMemcpy(dz, h_dz.data(), dz_n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(gin, 0, g_n * 4));
    CK(hipMemset(gin2, 0, g_n * 4));
    a.in = dz; a.scale = nullptr; a.shift = nullptr; a.res = nullptr; a.out = gin; a.leaky = 0; a.dgrad = 1;
    a.a_cs = Cout; a.Kc = (Cout + 31) & ~31; a.N = Cin; a.o_cs = Cin; a.r_cs = Cin;
    for (int t = 0; t < a.ntaps; ++t) {
      a.tap_dy[t] = (signed char)(k == 3 ? 1 - t / 3 : 0);
      a.tap_dx[t] = (signed char)(k == 3 ? 1 - t % 3 : 0);
    }
    a.w_split = wimg_d;
  }
  ConvArgs a2 = a;
  a2.out = dgrad ? gin2 : out2;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double K = (double)k * k * Cin, gflop = 2.0 * a.M * (double)Cout * K * 1e-9;
  float ms_e = 0, ms_s = 0;
  const bool split_only = getenv("VY_PROBE_SPLIT_ONLY") != nullptr;  // (power / clock sampling: one exact launch per round)
  for (int round = 0; round < 2; ++round) {  // interleaved: exact, split, exact, split
    for (int i = 0; i < 3; ++i) CK(vy_launch_conv_igemm(a, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < (split_only ? 1 : reps); ++i) CK(vy_launch_conv_igemm(a, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_e, e0, e1));
    for (int i = 0; i < 3; ++i) CK(vy_launch_conv_split(a2, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(vy_launch_conv_split(a2, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_s, e0, e1));
    const double us_e = ms_e * 1e3 / (split_only ? 1 : reps), us_s = ms_s * 1e3 / reps;
    int sbm, sbn, sks;
    vy_conv_split_cfg(a2, &sbm, &sbn, &sks);
    printf("[%dx%d k%d] ", sbm, sbn, sks);
    printf("conv B=%d H=%d Cin=%d Cout=%d k=%d s=%d res=%d | M=%d N=%d K=%.0f | exact %.1f us %.1f TF | split %.1f us %.1f TF-eq (%.0f TF bf16) | x%.2f\n",
           B, H, Cin, Cout, k, stride, res, a.M, a.N, K, us_e, gflop / us_e * 1e3, us_s, gflop / us_s * 1e3,
           6 * gflop / us_s * 1e3, us_e / us_s);
  }
  const size_t cmp_n = dgrad ? (size_t)B * (H + 2) * (H + 2) * Cin : out_n;
  std::vector<float> o1(cmp_n), o2(cmp_n);
  CK(hipMemcpy(o1.data(), dgrad ? gin : out, cmp_n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o2.data(), dgrad ? gin2 : out2, cmp_n * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0;
  size_t nbad = 0;
  for (size_t i = 0; i < cmp_n; ++i) {
    const double d = fabs((double)o1[i] - (double)o2[i]);
    if (!(d <= 1e30)) ++nbad;
    if (d > maxd) maxd = d;
    if (fabs(o1[i]) > maxv) maxv = fabs(o1[i]);
  }
  printf("  split vs exact: max |diff| %.3e  (max |value| %.3f, non-finite %zu)\n", maxd, maxv, nbad);
  if (dgrad) return 0;
  // float64 reference on sampled outputs
  double err_e = 0, err_s = 0;
  unsigned long long st = 424242ull;
  for (int smp = 0; smp < 4096; ++smp) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const int b = (int)((st >> 33) % B);
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const int y = (int)((st >> 33) % Ho);
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const int x = (int)((st >> 33) % Ho);
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const int n = (int)((st >> 33) % Cout);
    double s = 0;
    for (int t = 0; t < k * k; ++t) {
      const int dy = k == 3 ? t / 3 - 1 : 0, dx = k == 3 ? t % 3 - 1 : 0;
      const float* ap = &h_in[(((size_t)b * (H + 2) + y * stride + 1 + dy) * (H + 2) + x * stride + 1 + dx) * Cin];
      const float* wp = &h_w[((size_t)n * k * k + t) * Cin];
      for (int c = 0; c < Cin; ++c) s += (double)ap[c] * (double)wp[c];
    }
    double v = s * (double)h_sc[n] + (double)h_sh[n];
    v = v > 0 ? v : 0.1 * v;
    const size_t oi = (((size_t)b * (Ho + 2) + y + 1) * (Ho + 2) + x + 1) * Cout + n;
    if (res) v += (double)h_rs[oi];
    err_e = fmax(err_e, fabs(v - (double)o1[oi]));
    err_s = fmax(err_s, fabs(v - (double)o2[oi]));
  }
  printf("  max error vs float64 on 4096 samples: exact %.3e   split %.3e\n", err_e, err_s);
  return 0;
}
