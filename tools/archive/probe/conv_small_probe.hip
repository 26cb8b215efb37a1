// Times ONE conv launch shape through the small-tile kernel (conv_small.hip, 16x16x4 MFMA) and through the 32x32x2
// kernel, to find what bounds the small tile: built several times with -DVY_S16_ABLATE=<bits> (1 no LDS-DMA after the
// prologue, 2 no MFMAs, 4 no fragment reads, 8 no barriers).
//   for a in 0 1 2 4 8 3 6; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVY_S16_ABLATE=$a \
//       -I include -o conv_small_probe_$a tools/probe/conv_small_probe.hip; done
//   ./conv_small_probe_0 B H Cin Cout k      (tile via VY_CONV_FORCE=32x32 | 32x64 | 64x64 ...)
#include "../../videoyolo_amd/csrc/conv_igemm.hip"
#include "../../videoyolo_amd/csrc/conv_small.hip"

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

int main(int argc, char** argv) {
  if (argc < 6) {
    fprintf(stderr, "usage: %s B H Cin Cout k\n", argv[0]);
    return 2;
  }
  const int B = atoi(argv[1]), H = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]), k = atoi(argv[5]);
  const size_t in_n = (size_t)B * (H + 2) * (H + 2) * Cin, out_n = (size_t)B * (H + 2) * (H + 2) * Cout;
  const size_t w_n = (size_t)Cout * k * k * Cin;
  float *in, *out, *w, *sc, *sh;
  CK(hipMalloc(&in, in_n * 4));
  CK(hipMalloc(&out, out_n * 4));
  CK(hipMalloc(&w, w_n * 4));
  CK(hipMalloc(&sc, Cout * 4));
  CK(hipMalloc(&sh, Cout * 4));
  std::vector<float> h(in_n > w_n ? in_n : w_n);
  unsigned long long st = 88172645463325252ull;
  for (float& x : h) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    x = ((float)(st >> 40) * (1.0f / 16777216.0f) - 0.5f) * 0.2f;
  }
  CK(hipMemcpy(in, h.data(), in_n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), w_n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, h.data(), Cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sh, h.data(), Cout * 4, hipMemcpyHostToDevice));
  ConvArgs a;
  memset(&a, 0, sizeof a);
  a.in = in; a.w = w; a.scale = sc; a.shift = sh; a.res = nullptr; a.out = out;
  a.B = B; a.LH = H; a.LW = H; a.M = B * H * H;
  a.a_Hp = H + 2; a.a_Wp = H + 2; a.a_cs = Cin; a.a_co = 0; a.a_s = 1; a.a_oy = a.a_ox = 1;
  a.Kc = Cin; a.ntaps = k * k;
  for (int t = 0; t < a.ntaps; ++t) {
    a.tap_dy[t] = (signed char)(k == 3 ? t / 3 - 1 : 0);
    a.tap_dx[t] = (signed char)(k == 3 ? t % 3 - 1 : 0);
    a.tap_w[t] = (unsigned char)t;
  }
  a.w_taps = k * k; a.w_cin = Cin; a.w_cout = Cout; a.N = Cout;
  a.o_Hp = H + 2; a.o_Wp = H + 2; a.o_cs = Cout; a.o_co = 0; a.o_s = 1; a.o_oy = a.o_ox = 1; a.ups = 1;
  a.r_cs = Cout; a.r_co = 0; a.leaky = 1; a.dgrad = 0;
  int bm, bn;
  vy_conv_cfg(a, &bm, &bn);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) CK(vy_launch_conv_igemm(a, 0));
  CK(hipEventRecord(e0, 0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) CK(vy_launch_conv_igemm(a, 0));
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps, K = (double)k * k * Cin;
  const long long tiles = (long long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn);
  printf("ablate %d | B=%d H=%d Cin=%d Cout=%d k=%d | M=%d N=%d K=%.0f tile %dx%d tiles=%lld (%.2f per CU) | %7.1f us  %6.1f TF\n",
         VY_S16_ABLATE, B, H, Cin, Cout, k, a.M, a.N, K, bm, bn, tiles, tiles / 256.0, us, 2.0 * a.M * (double)Cout * K * 1e-3 / us);
  return 0;
}
