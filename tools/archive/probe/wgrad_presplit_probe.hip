// Split-fp32 weight gradient with PRE-SPLIT operands (three bf16 planes per activation plane, written once by a pre-pass,
// brought to LDS by LDS-DMA: no vector instruction in the k-loop) against the in-register split (wgrad_split.hip) and the
// exact kernel, on one layer shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-inline-asm -I include -o wgrad_presplit_probe \
//       tools/probe/wgrad_presplit_probe.hip videoyolo_amd/csrc/wgrad.hip && ./wgrad_presplit_probe B H Cin Cout k [stride=1] [splits=auto] [reps=10]
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

// pre-pass: fp32 plane [pixels][src_cs] (channels src_co .. src_co + C) -> three bf16 planes [pixels][C]
__global__ __launch_bounds__(256) void presplit_plane_kernel(const float* __restrict__ src, int src_cs, int src_co, int C,
                                                             long long pixels, unsigned char* __restrict__ dst,
                                                             long long plane_bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int cpp = C >> 3;  // 8-channel chunks per pixel
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= pixels * cpp) return;
  const long long p = i / cpp;
  const int c = (int)(i - p * cpp) * 8;
  const f32x4* s = reinterpret_cast<const f32x4*>(src + p * src_cs + src_co + c);
  vy_u32x4 H, M, L;
  ws_split8(s[0], s[1], H, M, L);
  unsigned char* d = dst + (p * C + c) * 2;
  *reinterpret_cast<vy_u32x4*>(d) = H;
  *reinterpret_cast<vy_u32x4*>(d + plane_bytes) = M;
  *reinterpret_cast<vy_u32x4*>(d + 2 * plane_bytes) = L;
#endif
}

#ifndef PS_NST
#define PS_NST 3
#endif

// zs / as: the three planes of dz ([B][Ho+2][Wo+2][Cout] bf16) and of the input ([B][a_Hp][a_Wp][Cin] bf16), plane strides
// in bytes.  a.tab was built for a_cs = Cin (compact input planes), a.a_cs = Cin, a.a_co = 0.
__global__ __launch_bounds__(256, 2) void wgrad_presplit_kernel(const WgradArgs a, const unsigned char* __restrict__ zs,
                                                                const unsigned char* __restrict__ as, const long long zs_plane,
                                                                const long long as_plane, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 128, KP = 16, NST = PS_NST;
  constexpr int ROW = 256;                   // bytes per pixel row of one plane: 128 bf16, 16-B chunk c of row q at c ^ ((q & 3) << 2)
  constexpr int PLANE = KP * ROW;            // 4096
  constexpr int IMG = 3 * PLANE;             // one operand of one stage
  constexpr int STAGE = 2 * IMG;             // A (dz) then B (input): 24 KiB
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, lrow = lane & 31;
  int tile_id = blockIdx.x, split = blockIdx.y;
  if (a.xcd_order) {
    const int gx = gridDim.x, nblk = gx * gridDim.y, L = blockIdx.y * gx + blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, idx = L >> 3;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    split = v / gx;
    tile_id = v - split * gx;
  }
  const int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;
  const int o0 = tile_m * BM, n0 = tile_n * BN;
  const int Ntot = a.k * a.k * a.Cin;
  const int p_begin = split * a.k_per_split;
  int p_end = p_begin + a.k_per_split;
  if (p_end > a.M) p_end = a.M;
  const int T = (p_end - p_begin + KP - 1) / KP;

  // DMA role: waves 0-1 bring the dz tile, waves 2-3 the input tile; a wave issues, per plane, the instructions of pixel
  // rows 8 (wave & 1) .. + 7 (two instructions of four 256-B rows).  Lane: row r = lane >> 4 of the instruction, LDS
  // chunk position lane & 15, which holds channel chunk (lane & 15) ^ (r << 2).
  const int op = wave >> 1;
  const int r4 = lane >> 4, chunk = (lane & 15) ^ (r4 << 2);
  const WsPix sb = ws_pixel_offsets(p_begin, a.Ho, a.Wo, a.z_cs, a.a_Hp, a.a_Wp, a.a_cs, a.stride);
  const unsigned char* g_base;
  long long g_plane;
  if (op == 0) {
    g_base = zs + (sb.zo >> 1) + (long long)(o0 + chunk * 8) * 2;
    g_plane = zs_plane;
  } else {
    const int bn = n0 + chunk * 8;
    const bool b_ok = bn < Ntot;
    const int tap = b_ok ? bn / a.Cin : 0;
    const int cin = b_ok ? bn - tap * a.Cin : 0;
    const int pad = a.k >> 1;
    const int dy = a.k == 3 ? tap / 3 - pad : 0, dx = a.k == 3 ? tap % 3 - pad : 0;
    g_base = as + (sb.ao >> 1) + ((long long)(dy * a.a_Wp + dx) * a.a_cs + cin) * 2;
    g_plane = as_plane;
  }
  const unsigned* tab = reinterpret_cast<const unsigned*>(a.tab + p_begin + 8 * (wave & 1) + r4) + op;  // .x (dz) or .y (input)
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const unsigned dma0 = lds0 + op * IMG + (wave & 1) * 8 * ROW;  // + stage * STAGE + plane * PLANE + instr * 4 * ROW

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // table entries as hand-written loads: every vector-memory instruction of the loop is counted by hand (vmcnt)
  unsigned e_cur[2], e_next[2];
  auto load_entries = [&](int t) {
    const unsigned* p = tab + (long long)t * KP * 2;
    asm volatile("global_load_dword %0, %1, off" : "=v"(e_next[0]) : "v"(p) : "memory");
    asm volatile("global_load_dword %0, %1, off offset:32" : "=v"(e_next[1]) : "v"(p) : "memory");  // + 4 pixel rows
  };
  auto dma = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned char* g = g_base + (e_cur[i] >> 1);
#pragma unroll
      for (int p = 0; p < 3; ++p)
        lds_dma16(reinterpret_cast<const float*>(g + p * g_plane), dma0 + stage * STAGE + p * PLANE + i * 4 * ROW);
    }
  };

  // transposed fragment reads (see wgrad_split_kernel): lane 4q+p of a 16-lane group addresses pixel row q, channels
  // 4p .. 4p+3 of the group's 16 channels
  const int g16 = lane >> 4, i16 = lane & 15;
  const int trow = 8 * (g16 >> 1) + (i16 >> 2);
  unsigned fa[2], fb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int cca = 8 * wm + 4 * i + 2 * (g16 & 1) + ((i16 & 3) >> 1), ccb = 8 * wn + 4 * i + 2 * (g16 & 1) + ((i16 & 3) >> 1);
    const int sw = (i16 >> 2) << 2;  // (row & 3) << 2
    fa[i] = lds0 + (unsigned)(trow * ROW + ((cca ^ sw) << 4) + (i16 & 1) * 8);
    fb[i] = lds0 + IMG + (unsigned)(trow * ROW + ((ccb ^ sw) << 4) + (i16 & 1) * 8);
  }
  typedef short ps_s16x4 __attribute__((ext_vector_type(4)));
  typedef short ps_s16x8 __attribute__((ext_vector_type(8)));
  auto tr8 = [&](unsigned addr) -> ws_bf16x8 {  // hipcc's builtin: the compiler schedules the reads against the MFMAs
    const ps_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps_s16x4*)(unsigned long long)addr);
    const ps_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps_s16x4*)(unsigned long long)(addr + 1024));
    const ps_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(ws_bf16x8, v);
  };
  auto compute = [&](int stage) {
    ws_bf16x8 af[3][2], bf[3][2];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[p][i] = tr8(fa[i] + stage * STAGE + p * PLANE);
        bf[p][i] = tr8(fb[i] + stage * STAGE + p * PLANE);
      }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]][i], bf[PB[t]][j], acc[i][j], 0, 0, 0);
  };

  // Order of a wave's vector-memory instructions: E(0) D(0) E(1)... ; iteration t issues E(t + NST) then D(t + NST - 1)
  // (6 instructions).  At the top of iteration t everything but the newest 6 (D(t + NST - 2)) ... must be complete.
  if (T > 0) {
    load_entries(0);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(e_next[0]), "+v"(e_next[1])::"memory");
    e_cur[0] = e_next[0], e_cur[1] = e_next[1];
    // prologue: stages 0 .. NST-2
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) {
      if (s < T) {
        if (s + 1 < T) load_entries(s + 1);
        dma(s);
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(e_next[0]), "+v"(e_next[1])::"memory");
        e_cur[0] = e_next[0], e_cur[1] = e_next[1];
      }
    }
    // here: D(0 .. NST-2) issued, e_cur = E(NST-1)
    int st_c = 0, st_d = NST - 1;
    for (int t = 0; t < T; ++t) {
      // D(t) complete: all but the newest (NST - 2) DMA groups
      if (NST == 3 && t + 1 < T) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();  // stage t written by every wave; stage (t - 1) free again: it is stage (t + NST - 1) % NST
      if (t + NST < T) load_entries(t + NST);
      if (t + NST - 1 < T) dma(st_d);
      compute(st_c);
      if (t + NST < T) {
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(e_next[0]), "+v"(e_next[1])::"memory");
        e_cur[0] = e_next[0], e_cur[1] = e_next[1];
      }
      st_c = st_c + 1 == NST ? 0 : st_c + 1;
      st_d = st_d + 1 == NST ? 0 : st_d + 1;
    }
  }

  float* slab = a.slabs + (long long)split * a.Cout * Ntot;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + lrow;
    if (n >= Ntot) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = o0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (o < a.Cout) slab[(long long)o * Ntot + n] = acc[i][j][r];
      }
  }
#endif
}

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
  unsigned char *zs, *as;
  void* tab;
  const size_t wn = (size_t)Cout * Ntot;
  const long long z_px = (long long)B * (Ho + 2) * (Ho + 2), a_px = (long long)B * (H + 2) * (H + 2);
  const long long zs_plane = z_px * Cout * 2, as_plane = a_px * Cin * 2;
  CK(hipMalloc(&d_a, h_a.size() * 4));
  CK(hipMalloc(&d_dz, h_dz.size() * 4));
  CK(hipMalloc(&zs, zs_plane * 3));
  CK(hipMalloc(&as, as_plane * 3));
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
  w.splits = splits; w.k_per_split = kps; w.xcd_order = 1;
  const int tiles_n = (Ntot + 127) / 128;
  auto pre = [&]() {
    hipLaunchKernelGGL(presplit_plane_kernel, dim3((unsigned)((z_px * (Cout / 8) + 255) / 256)), dim3(256), 0, 0, d_dz, Cout, 0, Cout,
                       z_px, zs, zs_plane);
    hipLaunchKernelGGL(presplit_plane_kernel, dim3((unsigned)((a_px * (Cin / 8) + 255) / 256)), dim3(256), 0, 0, d_a, Cin, 0, Cin,
                       a_px, as, as_plane);
  };
  auto run = [&]() {
    hipLaunchKernelGGL(wgrad_presplit_kernel, dim3(Cout / 128 * tiles_n, splits), dim3(256), 0, 0, w, zs, as, zs_plane, as_plane,
                       tiles_n);
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms_e = 0, ms_s = 0, ms_p = 0, ms_pp = 0;
  for (int round = 0; round < 2; ++round) {
    for (int i = 0; i < 2; ++i) CK(vy_launch_wgrad(w, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(vy_launch_wgrad(w, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_e, e0, e1));
    for (int i = 0; i < 2; ++i) CK(vy_launch_wgrad_split(w, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) CK(vy_launch_wgrad_split(w, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_s, e0, e1));
    CK(vy_launch_slab_reduce(slabs, splits, (long long)wn, dw1, 0));
    pre();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) pre();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_pp, e0, e1));
    for (int i = 0; i < 2; ++i) run();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    CK(hipEventElapsedTime(&ms_p, e0, e1));
    CK(vy_launch_slab_reduce(slabs, splits, (long long)wn, dw2, 0));
    const double gflop = 2.0 * M * (double)Cout * Ntot * 1e-9, us_e = ms_e * 1e3 / reps, us_s = ms_s * 1e3 / reps,
                 us_p = ms_p * 1e3 / reps, us_pp = ms_pp * 1e3 / reps;
    printf("wgrad B=%d H=%d Cin=%d Cout=%d k=%d s=%d | splits=%d x %d px | exact %.1f us | in-register split %.1f us %.1f TF-eq | "
           "pre-split %.1f us %.1f TF-eq + pre-pass %.1f us (%.0f GB/s) | x%.2f kernel, x%.2f with pre-pass\n",
           B, H, Cin, Cout, k, stride, splits, kps, us_e, us_s, gflop / us_s * 1e3, us_p, gflop / us_p * 1e3, us_pp,
           (double)(z_px * Cout + a_px * Cin) * 10.0 / us_pp * 1e-3, us_s / us_p, us_s / (us_p + us_pp));
  }
  std::vector<float> g1(wn), g2(wn);
  CK(hipMemcpy(g1.data(), dw1, wn * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(g2.data(), dw2, wn * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  double maxd = 0;
  for (size_t i = 0; i < wn; ++i) {
    if (memcmp(&g1[i], &g2[i], 4) != 0) ++bad;
    maxd = fmax(maxd, fabs((double)g1[i] - (double)g2[i]));
  }
  printf("  pre-split vs in-register split: %zu of %zu elements differ, max |diff| %.3e\n", bad, wn, maxd);
  return 0;
}
