// The PRODUCT's Winograd kernel (csrc/conv_wino.hip, through vy_launch_conv_wino) on one 3x3 stride-1 cell: launch time,
// error against a float64 evaluation of sampled outputs, and — built with -DVY_WINO_ABL=<bits> — the same launch with a
// part of its k-loop removed (what each part costs; results are garbage then and the check is skipped).
//   tools/probe/build_wino_abl.sh                    # builds wino_abl_probe and the ablation variants
//   ./wino_abl_probe B H Cin Cout [res=0] [reps=40]
#include "../../videoyolo_amd/csrc/conv_igemm.hip"
#include "../../videoyolo_amd/csrc/conv_split.hip"
#include "../../videoyolo_amd/csrc/conv_wino.hip"

#include <algorithm>
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
  if (argc < 5) return fprintf(stderr, "usage: %s B H Cin Cout [res=0] [reps=40]\n", argv[0]), 2;
  const int B = atoi(argv[1]), H = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]);
  const int res = argc > 5 ? atoi(argv[5]) : 0, reps = argc > 6 ? atoi(argv[6]) : 40;
  const int Wp = H + 2;
  const size_t in_n = (size_t)B * Wp * Wp * Cin, out_n = (size_t)B * Wp * Wp * Cout, w_n = (size_t)Cout * 9 * Cin;
  std::vector<float> h_in(in_n), h_w(w_n), h_sc(Cout), h_sh(Cout), h_rs(res ? out_n : 0);
  fill_normal(h_in, 1.0f, 88172645463325252ull);
  fill_normal(h_w, 1.0f / sqrtf(9.0f * Cin), 1234567ull);
  fill_normal(h_sc, 0.2f, 99ull);
  for (auto& v : h_sc) v += 1.0f;
  fill_normal(h_sh, 0.5f, 777ull);
  if (res) fill_normal(h_rs, 1.0f, 4242ull);
  if (getenv("VY_PROBE_ZERO")) {  // power experiment: all-zero operands (same instruction stream, no toggling)
    std::fill(h_in.begin(), h_in.end(), 0.f);
    std::fill(h_w.begin(), h_w.end(), 0.f);
  }
  if (getenv("VY_PROBE_SMALLW")) {  // ... and weights that are exact in bf16 (their m and l planes are zero: 3 of the 6 products multiply by zero)
    for (auto& v : h_w) v = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u);
  }
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < Wp; ++y)
      for (int x = 0; x < Wp; ++x)
        if (y == 0 || x == 0 || y == H + 1 || x == H + 1) memset(&h_in[(((size_t)b * Wp + y) * Wp + x) * Cin], 0, Cin * 4);
  float *in, *out, *w, *sc, *sh, *rs = nullptr;
  void* wimg;
  CK(hipMalloc(&in, in_n * 4));
  CK(hipMalloc(&out, out_n * 4));
  CK(hipMalloc(&w, w_n * 4));
  CK(hipMalloc(&sc, Cout * 4));
  CK(hipMalloc(&sh, Cout * 4));
  CK(hipMalloc(&wimg, vy_wino_weight_bytes(Cout, Cin)));
  CK(hipMemcpy(in, h_in.data(), in_n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h_w.data(), w_n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, h_sc.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, h_sh.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemset(out, 0, out_n * 4));
  if (res) {
    CK(hipMalloc(&rs, out_n * 4));
    CK(hipMemcpy(rs, h_rs.data(), out_n * 4, hipMemcpyHostToDevice));
  }
  CK(vy_launch_wino_weights(w, wimg, Cout, Cin, 0));
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.w = w; a.scale = sc; a.shift = sh; a.res = rs; a.out = out;
  a.B = B; a.LH = H; a.LW = H; a.M = B * H * H;
  a.a_Hp = Wp; a.a_Wp = Wp; a.a_cs = Cin; a.a_co = 0; a.a_s = 1; a.a_oy = a.a_ox = 1;
  a.Kc = Cin; a.ntaps = 9;
  for (int t = 0; t < 9; ++t) a.tap_dy[t] = (signed char)(t / 3 - 1), a.tap_dx[t] = (signed char)(t % 3 - 1), a.tap_w[t] = (unsigned char)t;
  a.w_taps = 9; a.w_cin = Cin; a.w_cout = Cout; a.N = Cout;
  a.o_Hp = Wp; a.o_Wp = Wp; a.o_cs = Cout; a.o_co = 0; a.o_s = 1; a.o_oy = a.o_ox = 1; a.ups = 1;
  a.r_cs = Cout; a.r_co = 0; a.leaky = 1; a.dgrad = 0;
  a.w_wino = wimg;
  if (!vy_conv_wino_supported(a)) return fprintf(stderr, "shape not supported by conv_wino\n"), 2;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  double best = 1e30;
  for (int round = 0; round < 3; ++round) {
    for (int i = 0; i < 3; ++i) CK(vy_launch_conv_wino(a, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(vy_launch_conv_wino(a, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fmin(best, ms * 1e3 / reps);
  }
  const double gflop = 2.0 * B * H * H * (double)Cout * 9.0 * Cin * 1e-9;
  printf("abl=%2d wino B=%d H=%d Cin=%d Cout=%d res=%d | %.1f us  %.1f TF-eq (%.0f TF bf16 issued)\n", (int)VY_WINO_ABL, B, H, Cin, Cout, res,
         best, gflop / best * 1e3, gflop / best * 1e3 * 4.0);
  if (VY_WINO_ABL) return 0;
  std::vector<float> h_out(out_n);
  CK(hipMemcpy(h_out.data(), out, out_n * 4, hipMemcpyDeviceToHost));
  double err = 0, maxv = 0;
  unsigned long long s2 = 424242ull;
  auto nxt = [&](int mod) {
    s2 = s2 * 6364136223846793005ull + 1442695040888963407ull;
    return (int)((s2 >> 33) % mod);
  };
  for (int smp = 0; smp < 4096; ++smp) {
    const int b = nxt(B), y = nxt(H), x = smp % 7 == 0 ? H - 1 : nxt(H), o = nxt(Cout);
    double s = 0;
    for (int dy = 0; dy < 3; ++dy)
      for (int dx = 0; dx < 3; ++dx)
        for (int c = 0; c < Cin; ++c)
          s += (double)h_in[(((size_t)b * Wp + y + dy) * Wp + x + dx) * Cin + c] * (double)h_w[(((size_t)o * 3 + dy) * 3 + dx) * Cin + c];
    double v = s * (double)h_sc[o] + (double)h_sh[o];
    v = v > 0.1 * v ? v : 0.1 * v;
    const size_t oi = (((size_t)b * Wp + y + 1) * Wp + x + 1) * Cout + o;
    if (res) v += h_rs[oi];
    err = fmax(err, fabs((double)h_out[oi] - v));
    maxv = fmax(maxv, fabs(v));
  }
  // the zero frame must still be zero
  double border = 0;
  for (int b = 0; b < B; b += (B > 4 ? B / 4 : 1))
    for (int x = 0; x < Wp; ++x)
      for (int c = 0; c < Cout; c += 7) {
        border = fmax(border, fabs((double)h_out[(((size_t)b * Wp + 0) * Wp + x) * Cout + c]));
        border = fmax(border, fabs((double)h_out[(((size_t)b * Wp + H + 1) * Wp + x) * Cout + c]));
        border = fmax(border, fabs((double)h_out[(((size_t)b * Wp + x) * Wp + 0) * Cout + c]));
        border = fmax(border, fabs((double)h_out[(((size_t)b * Wp + x) * Wp + H + 1) * Cout + c]));
      }
  printf("  max error vs float64 on 4096 samples: %.3e (max |value| %.3f); border max %.1e\n", err, maxv, border);
  return err < 1e-4 && border == 0 ? 0 : 1;
}
