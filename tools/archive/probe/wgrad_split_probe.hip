// Split-fp32 weight gradient against the exact fp32 one on one layer shape: speed and distance.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-inline-asm -I include -o wgrad_split_probe \
//       tools/probe/wgrad_split_probe.hip videoyolo_amd/csrc/wgrad.hip && ./wgrad_split_probe B H Cin Cout k [stride=1] [splits=auto] [reps=10]
#include "../../videoyolo_amd/csrc/wgrad_split.hip"

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

static void fill_plane(std::vector<float>& h, int B, int H, int W, int C, unsigned long long seed) {
  unsigned long long st = seed;
  for (int b = 0; b < B; ++b)
    for (int y = 0; y < H + 2; ++y)
      for (int x = 0; x < W + 2; ++x)
        for (int c = 0; c < C; ++c) {
          float acc = 0.f;
          for (int j = 0; j < 4; ++j) {
            st ^= st << 13;
            st ^= st >> 7;
            st ^= st << 17;
            acc += (float)(st >> 40) * (1.0f / 16777216.0f) - 0.5f;
          }
          const bool border = y == 0 || x == 0 || y == H + 1 || x == W + 1;
          h[(((size_t)b * (H + 2) + y) * (W + 2) + x) * C + c] = border ? 0.f : acc * 1.7320508f;
        }
}

int main(int argc, char** argv) {
  if (argc < 6) return fprintf(stderr, "usage: %s B H Cin Cout k [stride=1] [splits=0:auto] [reps=10]\n", argv[0]), 2;
  const int B = atoi(argv[1]), H = atoi(argv[2]), Cin = atoi(argv[3]), Cout = atoi(argv[4]), k = atoi(argv[5]);
  const int stride = argc > 6 ? atoi(argv[6]) : 1;
  int splits = argc > 7 ? atoi(argv[7]) : 0;
  const int reps = argc > 8 ? atoi(argv[8]) : 10;
  const int Ho = H / stride;
  const int M = B * Ho * Ho, Ntot = k * k * Cin;
  if (splits <= 0) {  // about 1024 blocks
    const int tiles = ((Cout + 127) / 128) * ((Ntot + 127) / 128);
    splits = std::max(1, 1024 / tiles);
  }
  int kps = ((M + splits - 1) / splits + 31) / 32 * 32;
  splits = (M + kps - 1) / kps;
  std::vector<float> h_a((size_t)B * (H + 2) * (H + 2) * Cin), h_dz((size_t)B * (Ho + 2) * (Ho + 2) * Cout);
  fill_plane(h_a, B, H, H, Cin, 88172645463325252ull);
  fill_plane(h_dz, B, Ho, Ho, Cout, 1234567ull);
  float *d_a, *d_dz, *slabs, *dw1, *dw2;
  void* tab;
  const size_t wn = (size_t)Cout * Ntot;
  CK(hipMalloc(&d_a, h_a.size() * 4));
  CK(hipMalloc(&d_dz, h_dz.size() * 4));
  CK(hipMalloc(&slabs, (size_t)splits * wn * 4));
  CK(hipMalloc(&dw1, wn * 4));
  CK(hipMalloc(&dw2, wn * 4));
  CK(hipMalloc(&tab, vy_wgrad_table_entries(M) * 8));
  CK(hipMemcpy(d_a, h_a.data(), h_a.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_dz, h_dz.data(), h_dz.size() * 4, hipMemcpyHostToDevice));
  CK(vy_launch_wgrad_table(tab, M, (int)vy_wgrad_table_entries(M), Ho, Ho, Cout, H + 2, H + 2, Cin, stride, B, kps, 0));
  WgradArgs w;
  memset(&w, 0, sizeof w);
  w.dz = d_dz; w.a = d_a; w.slabs = slabs; w.zero = nullptr; w.tab = (const uint2*)tab;
  w.B = B; w.Ho = Ho; w.Wo = Ho; w.M = M; w.z_cs = Cout; w.Cout = Cout;
  w.a_Hp = H + 2; w.a_Wp = H + 2; w.a_cs = Cin; w.a_co = 0; w.stride = stride; w.k = k; w.Cin = Cin;
  w.splits = splits; w.k_per_split = kps;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms_e = 0, ms_s = 0;
  for (int round = 0; round < 2; ++round) {
    for (int i = 0; i < 2; ++i) CK(vy_launch_wgrad(w, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(vy_launch_wgrad(w, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_e, e0, e1));
    CK(vy_launch_slab_reduce(slabs, splits, (long long)wn, dw1, 0));
    for (int i = 0; i < 2; ++i) CK(vy_launch_wgrad_split(w, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(vy_launch_wgrad_split(w, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_s, e0, e1));
    CK(vy_launch_slab_reduce(slabs, splits, (long long)wn, dw2, 0));
    const double gflop = 2.0 * M * (double)Cout * Ntot * 1e-9, us_e = ms_e * 1e3 / reps, us_s = ms_s * 1e3 / reps;
    printf("wgrad B=%d H=%d Cin=%d Cout=%d k=%d s=%d | M=%d Ntot=%d splits=%d x %d px | exact %.1f us %.1f TF | split %.1f us %.1f TF-eq | x%.2f\n",
           B, H, Cin, Cout, k, stride, M, Ntot, splits, kps, us_e, gflop / us_e * 1e3, us_s, gflop / us_s * 1e3, us_e / us_s);
  }
  std::vector<float> g1(wn), g2(wn);
  CK(hipMemcpy(g1.data(), dw1, wn * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(g2.data(), dw2, wn * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0;
  for (size_t i = 0; i < wn; ++i) {
    maxd = fmax(maxd, fabs((double)g1[i] - (double)g2[i]));
    maxv = fmax(maxv, fabs((double)g1[i]));
  }
  // float64 reference on sampled weight elements
  double err_e = 0, err_s = 0;
  unsigned long long st = 424242ull;
  for (int smp = 0; smp < 512; ++smp) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const int o = (int)((st >> 33) % Cout);
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    const int n = (int)((st >> 33) % Ntot);
    const int tap = n / Cin, c = n % Cin, dy = k == 3 ? tap / 3 - 1 : 0, dx = k == 3 ? tap % 3 - 1 : 0;
    double s = 0;
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < Ho; ++y)
        for (int x = 0; x < Ho; ++x)
          s += (double)h_dz[(((size_t)b * (Ho + 2) + y + 1) * (Ho + 2) + x + 1) * Cout + o] *
               (double)h_a[(((size_t)b * (H + 2) + y * stride + 1 + dy) * (H + 2) + x * stride + 1 + dx) * Cin + c];
    err_e = fmax(err_e, fabs(s - (double)g1[(size_t)o * Ntot + n]));
    err_s = fmax(err_s, fabs(s - (double)g2[(size_t)o * Ntot + n]));
  }
  printf("  split vs exact: max |diff| %.3e (max |value| %.3f) | error vs float64 on 512 samples: exact %.3e  split %.3e\n", maxd,
         maxv, err_e, err_s);
  return 0;
}
