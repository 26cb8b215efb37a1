// wgrad_split.hip — OPT-IN split-fp32 weight gradient on the bf16 matrix core (conv mode
// VY_CONV_SPLIT_BF16X3_TRAIN): the same arithmetic as conv_split.hip (every fp32 operand cut exactly into three bf16
// numbers, six partial products per multiply, fp32 accumulate), for the reduction the training step is bound by
// (wgrad.hip: 17.6 ms of kernel time per 30.4 ms step at 416x416, batch 16).  Not the parity path.
//
// Replaces the weight-gradient half of mxnet's Convolution backward (train_yolov3.py:631 through
// models/definitions/layers.py:66):   dW[o][tap][cin] = sum_p dz[p][o] * a[p*s + tap][cin]
// GEMM view:  D[o][n] = sum_k A[o][k] * B[k][n],  k = output pixel p,  n = (tap, cin);  split-K over pixel ranges into
// slabs [split][Cout][taps*Cin], summed in order by vy_launch_slab_reduce (wgrad.hip) — same planner, same pixel table.
//
// Both operands are activations, pixel-major in memory (NHWC planes), and BOTH are split in registers on the way into
// LDS: per k-step of 16 pixels a thread loads 8 dz channels and 8 input channels of one pixel, cuts them into the three
// planes (2 x 44 vector instructions) and stores 6 x 16 B.  The MFMA fragments need 8 consecutive k (pixels) of one
// channel: `ds_read_b64_tr_b16` (through hipcc's builtin) delivers a 4-pixel x 16-channel block column-major, two reads per
// fragment.  LDS image
// per operand and stage: [plane][16 pixels][128 channels + 32 pad] bf16 — the 320-B row pitch puts the four pixel rows
// of a transposed read on disjoint banks.  Two stages (60 KiB): two blocks per CU.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "conv_device.h"

#if defined(__HIP_DEVICE_COMPILE__)
typedef __bf16 ws_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned ws_cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ void ws_split8(const f32x4 v0, const f32x4 v1, vy_u32x4& H, vy_u32x4& M, vy_u32x4& L) {
  const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = x[2 * j], x1 = x[2 * j + 1];
    const unsigned h = ws_cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const unsigned m = ws_cvt_pk_bf16(r0, r1);
    const float l0 = r0 - __builtin_bit_cast(float, m << 16), l1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    H[j] = h;
    M[j] = m;
    L[j] = ws_cvt_pk_bf16(l0, l1);
  }
}
struct WsPix {
  unsigned long long zo, ao;
};
__device__ __forceinline__ WsPix ws_pixel_offsets(int p, int Ho, int Wo, int z_cs, int a_Hp, int a_Wp, int a_cs, int stride) {
  const int x = p % Wo, t = p / Wo, y = t % Ho, b = t / Ho;
  WsPix o;
  o.zo = ((unsigned long long)(b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * z_cs * 4ull;
  o.ao = ((unsigned long long)(b * a_Hp + y * stride + 1) * a_Wp + x * stride + 1) * a_cs * 4ull;
  return o;
}
#endif

__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(const WgradArgs a, const int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 128, KP = 16;
  constexpr int PITCH = 320;                 // bytes per pixel row of one plane: 128 bf16 + 64 B pad
  constexpr int PLANE = KP * PITCH;          // 5120
  constexpr int IMG = 3 * PLANE;             // one operand of one stage
  constexpr int STAGE = 2 * IMG;             // A (dz) then B (input)
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, lrow = lane & 31;
  int tile_id = blockIdx.x, split = blockIdx.y;
  if (a.xcd_order) {  // an XCD works through a contiguous run of (split, tile) pairs (wgrad.hip)
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

  // staging role of this thread: pixel row `pix` of the k-step, 8-channel chunk `chunk` of both tiles
  const int pix = tid >> 4, chunk = tid & 15;
  const int oc = o0 + chunk * 8;                       // dz channels oc .. oc+7 (Cout % 128 == 0: always valid)
  const int bn = n0 + chunk * 8;                       // n columns bn .. bn+7: one tap (Cin % 8 == 0)
  const bool b_ok = bn < Ntot;                         // columns past taps*Cin: tap 0 / channel 0, never stored
  const int tap = b_ok ? bn / a.Cin : 0;
  const int cin = b_ok ? bn - tap * a.Cin : 0;
  const int pad = a.k >> 1;
  const int dy = a.k == 3 ? tap / 3 - pad : 0, dx = a.k == 3 ? tap % 3 - pad : 0;
  const WsPix sb = ws_pixel_offsets(p_begin, a.Ho, a.Wo, a.z_cs, a.a_Hp, a.a_Wp, a.a_cs, a.stride);
  const unsigned char* z_base = reinterpret_cast<const unsigned char*>(a.dz) + sb.zo + (long long)oc * 4;
  const unsigned char* a_base = reinterpret_cast<const unsigned char*>(a.a) + sb.ao +
                                ((long long)a.a_co + (long long)(dy * a.a_Wp + dx) * a.a_cs + cin) * 4;
  const uint2* tab = a.tab + p_begin + pix;
  const unsigned st_off = (unsigned)(pix * PITCH + chunk * 16);  // this thread's 16 B inside a plane

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  f32x4 za[2], aa[2];
  auto load = [&](const uint2 e) {
    const f32x4* pz = reinterpret_cast<const f32x4*>(z_base + e.x);
    const f32x4* pa = reinterpret_cast<const f32x4*>(a_base + e.y);
    za[0] = pz[0];
    za[1] = pz[1];
    aa[0] = pa[0];
    aa[1] = pa[1];
  };
  auto store = [&](int stage) {
    vy_u32x4 H, M, L;
    unsigned char* d = smem + stage * STAGE + st_off;
    ws_split8(za[0], za[1], H, M, L);
    *reinterpret_cast<vy_u32x4*>(d) = H;
    *reinterpret_cast<vy_u32x4*>(d + PLANE) = M;
    *reinterpret_cast<vy_u32x4*>(d + 2 * PLANE) = L;
    ws_split8(aa[0], aa[1], H, M, L);
    *reinterpret_cast<vy_u32x4*>(d + IMG) = H;
    *reinterpret_cast<vy_u32x4*>(d + IMG + PLANE) = M;
    *reinterpret_cast<vy_u32x4*>(d + IMG + 2 * PLANE) = L;
  };
  // transposed fragment reads: lane 4q+p of a 16-lane group addresses pixel row q, channels 4p .. 4p+3 of the group's
  // 16 channels; lanes 0-15 / 16-31 take channels 0-15 / 16-31 of a 32-channel MFMA tile at pixels 0-3, lanes 32-63 the
  // same at pixels 8-11 (the second read: + 4 pixels).  Lane l then holds A[channel l & 31][k = 8 (l >> 5) + j].
  const int g16 = lane >> 4, i16 = lane & 15;
  const unsigned tr_off = (unsigned)((8 * (g16 >> 1) + (i16 >> 2)) * PITCH + (16 * (g16 & 1) + 4 * (i16 & 3)) * 2);
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const unsigned fa0 = lds0 + tr_off + (unsigned)(wm * 64 * 2);         // A tile of this wave: channels wm*64 ..
  const unsigned fb0 = lds0 + IMG + tr_off + (unsigned)(wn * 64 * 2);   // B tile: columns wn*64 ..
  typedef short ws_s16x4 __attribute__((ext_vector_type(4)));
  typedef short ws_s16x8 __attribute__((ext_vector_type(8)));
  auto tr8 = [&](unsigned addr) -> ws_bf16x8 {
    const ws_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws_s16x4*)(unsigned long long)addr);
    const ws_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws_s16x4*)(unsigned long long)(addr + 1280));
    const ws_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(ws_bf16x8, v);
  };
  auto compute = [&](int stage) {
    ws_bf16x8 af[3][2], bf[3][2];
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};  // (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
    auto prod = [&](int t) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA[t]][i], bf[PB[t]][j], acc[i][j], 0, 0, 0);
    };
    // The transposed reads go through the compiler's builtin, so that hipcc counts them and interleaves them with the
    // MFMAs itself.  As inline asm with ONE hand-placed wait before the first MFMA the kernel was 10 % slower (166 against
    // 150 us on the 52x52 128->256 layer), with hand-staged waits per product group 4 % slower (155 us); finer stages, or
    // the reads issued ahead of the next k-step's split arithmetic, slower again (tools/probe/run_wgrad_staged.sh).
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[p][i] = tr8(fa0 + stage * STAGE + p * PLANE + i * 64);
        bf[p][i] = tr8(fb0 + stage * STAGE + p * PLANE + i * 64);
      }
#pragma unroll
    for (int t = 0; t < 6; ++t) prod(t);
  };

  if (T > 0) {
    uint2 e_next = tab[0];                       // entry of k-step 0
    load(e_next);
    if (T > 1) e_next = tab[KP];                 // entry of k-step 1
    store(0);
    if (T > 1) load(e_next);
    if (T > 2) e_next = tab[2 * KP];
    for (int t = 0; t < T; ++t) {
      lds_barrier();                             // stage t & 1 written by every thread; stage (t + 1) & 1 free again
      if (t + 1 < T) store((t + 1) & 1);         // k-step t + 1: its loads were issued a whole k-step ago
      if (t + 2 < T) load(e_next);
      if (t + 3 < T) e_next = tab[(t + 3) * KP];
      compute(t & 1);
    }
  }

  // D[o][n]: C/D map of the 32x32 MFMA: column = lane & 31 (n), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (o)
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

bool vy_wgrad_split_supported(const WgradArgs& a) {
  return a.Cout % 128 == 0 && a.Cin % 32 == 0 && a.k_per_split % 32 == 0 && a.splits >= 1 && (a.z_cs & 3) == 0 &&
         (a.a_cs & 3) == 0 && (a.a_co & 3) == 0 && a.tab != nullptr;
}

hipError_t vy_launch_wgrad_split(const WgradArgs& a_in, hipStream_t s) {
  WgradArgs a = a_in;
  if (!vy_wgrad_split_supported(a)) return hipErrorInvalidValue;
  static const int xcd_order = getenv("VY_WGRAD_XCD") ? atoi(getenv("VY_WGRAD_XCD")) : 1;
  a.xcd_order = xcd_order;
  const int Ntot = a.k * a.k * a.Cin;
  const int tiles_n = (Ntot + 127) / 128;
  hipLaunchKernelGGL(wgrad_split_kernel, dim3(a.Cout / 128 * tiles_n, a.splits), dim3(256), 0, s, a, tiles_n);
  return hipGetLastError();
}
