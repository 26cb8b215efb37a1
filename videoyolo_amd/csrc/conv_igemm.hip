// conv_igemm.hip — 3x3 / 1x1 convolution (forward and data-gradient) as an implicit GEMM on the
// gfx950 fp32 matrix core.
//
// Replaces the mxnet operator chain behind the reference's `_conv2d` cell
// (models/definitions/layers.py:63-70: Conv2D(no bias) -> BatchNorm -> LeakyReLU(0.1)), the
// residual add of DarknetBasicBlockV3 (darknet/three_darknet.py:119-123), the 1x1 prediction conv
// with bias (yolo/yolo3.py:62) and `_upsample` + concat (layers.py:11-20, yolo3.py:1167-1177), all
// in ONE kernel: the affine / activation / residual / x2-replicate happen on the accumulator.
// In training the same kernel produces the raw conv output plus per-tile channel sums for the
// batch statistics, and (dgrad mode) the gradient w.r.t. the conv input that mxnet's autograd
// computes for Convolution (train_yolov3.py:631).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * W[n][k]
//   m = (b, y, x) pixel of a logical grid   M = B*LH*LW
//   n = output channel                      N
//   k = (tap, channel)                      K = ntaps*Kc, Kc % 32 == 0
// A is never materialised: activations are zero-bordered NHWC planes (kernels.h), so the A row of
// pixel m for tap t is 32 contiguous floats at a fixed offset from the pixel's centre address.
//
// Numerics: v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain in k order (k0 from lanes 0-31, k1
// from lanes 32-63).  Each lane fetches 4 consecutive channels with one ds_read_b128; the lower
// half-wave takes channels 8g..8g+3 and the upper half 8g+4..8g+7 of each group of 8, so every
// output element is ONE fma chain over taps (kh,kw) ascending and, inside a tap, channels in the
// order 0,4,1,5,2,6,3,7 per aligned group of 8 — the order the CPU checker uses too, which makes
// the conv stack bit-reproducible with zero cross-lane traffic.
//
// Tiling (MI355X): 4 waves, tiles 128 x {128, 64, 32} and 64 x 64; k-step 32; per k-step the A and W
// tiles (128 B per row) are brought in by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
// instruction) into a double buffer; the 16-B chunk index is XOR-swizzled on the SOURCE side with
// (row>>1)&7 so the ds_read_b128 of 32 different rows at one k-chunk is bank-conflict free.
// 128x128 tile: 66 KiB of LDS -> 2 blocks / CU, 64 accumulator VGPRs / lane.
// dgrad reads W as the [k][n] operand: its LDS tile is 32 k-rows of BN contiguous floats, read with
// ds_read_b32 (32 consecutive floats per half-wave: conflict-free without a swizzle).
#include <atomic>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "../../include/vy_math.h"

#include "conv_device.h"
#include "sk_schedule.h"
#include "conv_cost_model.h"

// Stream-K hand-off between the block that starts a tile and the block that finishes it (SK instances only).
// `tiles` tiles in the launch (the grid is smaller); partials: [grid][BM * BN] floats; flags: [grid] words, zero between
// launches (the consumer clears the flag it waited for).
struct SkArgs {
  float* partials;
  unsigned* flags;
  int tiles;
  // > 0: split-K launch (grid = tiles x ksplit): block (tile, chunk c) runs k-steps [c T / S, (c + 1) T / S) as its OWN
  // chain from +0; chunks 0 .. S - 2 leave their sums in a slab, the block of chunk S - 1 adds them IN CHUNK ORDER to its
  // own — ((P0 + P1) + P2) + P3 — and runs the epilogue.  0: the stream-K schedule.
  int ksplit;
};

// One tile of the implicit GEMM, k-steps [kb, ke): from zero or (SK) from the previous block's partial sums, to the
// epilogue or (SK) to this block's partial slab.  `Args` is ConvArgs in the address space the caller reads it from.
template <int BM, int BN, int WM, int WN, bool DGRAD, int NS, bool SK, bool CK, typename Args>
__device__ __forceinline__ void conv_tile(Args& a, const int tiles_n, const SkArgs& sk, unsigned char* smem,
                                          const int vblk, const int v, const int kb, const int ke,
                                          const bool load_partial, const bool store_partial, const int ks_reduce = 0,
                                          const int ch_len = 0, const int n_tiles = 0) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NW = WM * WN, NT = NW * 64;               // waves / threads per block
  constexpr int A_INSTR = BM / 8 / NW, B_INSTR = BN / 8 / NW;  // LDS-DMA instructions per wave per k-step
  static_assert(NW == 4 || NW == 8, "4 or 8 waves");
  static_assert(A_INSTR >= 1 && B_INSTR >= 1, "tile too small for the wave count");
  static_assert(TM >= 1 && TN >= 1, "tile");
  long long* in_off = reinterpret_cast<long long*>(smem + NS * STAGE);
  // epilogue row tables: byte offset of the row's output pixel (and of its addend pixel) relative to the
  // tile's first pixel; kInvalidRow for rows past M.  The epilogue addresses memory through buffer
  // descriptors based at the tile's first pixel: 32-bit offsets (one v_add per element instead of
  // 64-bit multiply-adds), and an offset with bit 31 set is out of range for the descriptor, so the
  // hardware drops the store / returns 0 for the load — no per-element branches.
  unsigned* o_off = reinterpret_cast<unsigned*>(in_off + BM);
  unsigned* r_off = o_off + BM;
  constexpr unsigned kInvalidRow = 0x80000000u;

  int tid_ = threadIdx.x;
  // SK: the tile body runs in a loop over work items.  Everything derived from the thread index is loop-invariant, and
  // hipcc hoists it all out of the loop and keeps it live (fragment addresses, swizzles, epilogue columns: 78 -> 140
  // VGPRs for the 64x64 tile, spills for 128x128, an occupancy step lost).  Opaque per item, it is recomputed instead.
  if (SK) asm volatile("" : "+v"(tid_));
  const int tid = tid_;
#ifdef VY_CONV_TRACE  // tools/probe/conv_tile_trace.hip: thread 0 stamps the tile's phases (never compiled into the library)
#define VY_TRACE(slot) \
  if (a.trace && tid == 0) a.trace[(long long)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();
#else
#define VY_TRACE(slot)
#endif
  VY_TRACE(0)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5;
  const int cchunks = a.Kc >> 5;

  if (SK) __syncthreads();  // the previous item's epilogue is done with the row tables
  const int tile_m = v / tiles_n, tile_n = v - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // output pixel of the tile's first row (m0 < M always): wave-uniform
  long long pix0;
  {
    const int t = (int)fd_div((unsigned)m0, a.fd_lw);
    const int x = m0 - t * a.LW;
    const int b = (int)fd_div((unsigned)t, a.fd_lh);
    const int y = t - b * a.LH;
    const long long p = (long long)(b * a.o_Hp + y * a.o_s + a.o_oy) * a.o_Wp + x * a.o_s + a.o_ox;
    pix0 = ((long long)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)p);
  }
  for (int rr = tid; rr < BM; rr += NT) {
    const int m = m0 + rr;
    const int mm = m < a.M ? m : a.M - 1;
    // (two multiply-highs instead of two emulated 32-bit divisions per row: the prologue is on the critical path of
    // the short launches at small batches)
    const int t = (int)fd_div((unsigned)mm, a.fd_lw);
    const int x = mm - t * a.LW;
    const int b = (int)fd_div((unsigned)t, a.fd_lh);
    const int y = t - b * a.LH;
    in_off[rr] = ((long long)(b * a.a_Hp + y * a.a_s + a.a_oy) * a.a_Wp + x * a.a_s + a.a_ox) * a.a_cs + a.a_co;
    const unsigned rel = (unsigned)(((long long)(b * a.o_Hp + y * a.o_s + a.o_oy) * a.o_Wp + x * a.o_s + a.o_ox) - pix0);
    unsigned oo_row = (m < a.M) ? rel * (unsigned)a.o_cs * 4u : kInvalidRow;
    // x2-replicated store (nearest upsample fused, layers.py:11-20) cropped to the route's size (slice_like,
    // yolo3.py:1177): bit 0 / bit 1 of the (4-byte aligned) offset drop the row's second column / second row
    if (a.ups == 2 && m < a.M) oo_row |= (2 * x + 1 >= a.o_Wp - 2 ? 1u : 0u) | (2 * y + 1 >= a.o_Hp - 2 ? 2u : 0u);
    o_off[rr] = oo_row;
    r_off[rr] = (m < a.M) ? rel * (unsigned)a.r_cs * 4u : kInvalidRow;
  }
  __syncthreads();

  // LDS-DMA sources: wave-uniform base (advanced per k-step by the uniform tap / k offset) + per-lane 32-bit byte
  // offset relative to the tile's first row (rows ascend in memory, a tile spans less than one image plane)
  unsigned a_voff[A_INSTR];
  unsigned b_voff[B_INSTR];
  const long long in_off0 = in_off[0];
  const float* a_sbase;
  {
    const unsigned long long p = (unsigned long long)(a.in + in_off0);
    a_sbase = (const float*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
                             (unsigned)__builtin_amdgcn_readfirstlane((int)p));
  }
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int row = (j * NW + wave) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_voff[j] = (unsigned)((in_off[row] - in_off0) * 4 + chunk * 16);
  }
  const int wK = a.w_taps * a.w_cin;  // floats per weight row (one cout)
  int b_krow[B_INSTR];                // dgrad: k-row of this lane inside the 32-row W tile
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    if (!DGRAD) {
      const int row = (j * NW + wave) * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      int n = n0 + row;
      n = n < a.N ? n : a.N - 1;
      b_voff[j] = (unsigned)((n - n0) * wK * 4 + chunk * 16);  // relative to the tile's first weight row
      b_krow[j] = 0;
    } else {
      constexpr int CPR = BN / 4;    // 16-B chunks per k-row
      constexpr int RPI = 64 / CPR;  // k-rows per wave instruction
      const int row = (j * NW + wave) * RPI + lane / CPR;
      int n = n0 + (lane % CPR) * 4;
      n = n < a.N ? n : 0;
      (void)n;
      b_voff[j] = 0;  // set below (needs the tap / channel-chunk geometry)
      b_krow[j] = row;
    }
  }

  f32x16 acc[TM][TN];
  if constexpr (SK) {
    // The head of a continued tile was computed by block vblk - 1 (right after its whole-tile waves) and stored WRITE-THROUGH (sc1).
    // One lane polls its flag, the block's barrier, then sc1 loads (they bypass this CU's L1; no agent-scope acquire: the
    // slab is the only memory another block of this launch writes, and every load of it is one of these).  The loads are
    // issued for EVERY item, through a buffer descriptor whose range is 0 unless this item continues a tile:
    // out-of-range loads return 0.0f without touching memory — the accumulators have ONE definition (a branch here
    // made hipcc keep two copies of them: 2x the registers).
    if (load_partial && tid == 0) {
      // bounded: the predecessor was dispatched before this block and reaches its HEAD without waiting for anything, so
      // the flag normally goes up within microseconds; two seconds (s_memrealtime ticks at 100 MHz) without it means the
      // launch is broken (a faulted predecessor, a workspace shared by two streams): abort the kernel instead of hanging
      // the queue — the host sees hipErrorLaunchFailure at its next synchronisation
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (__hip_atomic_load(sk.flags + (vblk - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) __builtin_trap();
      }
    }
    if (load_partial) __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: keeps the loads below the poll
    const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc(
        sk.partials + (size_t)(vblk > 0 ? vblk - 1 : 0) * (BM * BN), 0, load_partial ? BM * BN * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v4 = buf_load_f32x4_sc1(slab, (unsigned)((((i * TN + j) * 4 + q) * NT + tid) * 16));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][q * 4 + e] = v4[e];
        }
    // (the flag is cleared for the next launch once every thread has its values: after the k-loop, see below)
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }

  // LDS byte address of smem (a 32-bit LDS pointer), wave-uniform
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const int T = ke - kb;

  const int lrow = lane & 31;
  // ---- k-loop.  Wave-uniform tile state (tap, channel chunk -> A / W offsets) is advanced once per tile
  // with scalar arithmetic; the last tile is peeled (no prefetch); fragments are double-buffered in
  // registers; sched_barriers pin the order  [MFMA steps 0,1] [a third of the next tile's DMA] [step 2]
  // [next group's ds_reads] [step 3]  per 8-channel group.
  int n_tap = SK ? kb / cchunks : 0, n_cc = SK ? kb - (kb / cchunks) * cchunks : 0;
  int a_koff = 0;          // floats; |(dy * Wp + dx) * cs| < 2^23: 32-bit scalar arithmetic
  long long b_koff = 0;
  bool n_lastcc = false;
  // forward: the tile's first weight row; dgrad: the tensor itself (kernel argument + uniform values: scalar)
  const float* b_sbase = DGRAD ? a.w : a.w + (long long)n0 * wK;
  auto advance = [&]() {
    const int tdy = (int)((a.pk_dy >> (2 * n_tap)) & 3u) - 1, tdx = (int)((a.pk_dx >> (2 * n_tap)) & 3u) - 1;
    const int tw = (int)((a.pk_w >> (4 * n_tap)) & 15ull);
    a_koff = (tdy * a.a_Wp + tdx) * a.a_cs + n_cc * 32;
    if (!DGRAD)
      b_koff = tw * a.w_cin + n_cc * 32;
    else
      b_koff = ((long long)n_cc * 32 * a.w_taps + tw) * a.w_cin;
    n_lastcc = (n_cc == cchunks - 1);
    if (++n_cc == cchunks) {
      n_cc = 0;
      ++n_tap;
    }
  };
  // dgrad: W is the [k = cout][n = cin] operand; lane offset = its k-row and n chunk, constant over the k-loop
  // except in the last channel chunk of a cout that is not a multiple of 32 (the prediction convs: rows past
  // cout re-read the last one) — a second offset set, chosen per DMA by a select on a scalar condition
  unsigned b_voff_last[B_INSTR];
  if (DGRAD) {
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      constexpr int CPR = BN / 4;
      int n = n0 + (lane % CPR) * 4;
      n = n < a.N ? n : 0;
      b_voff[j] = (unsigned)((n + b_krow[j] * a.w_taps * a.w_cin) * 4);
      int o = (cchunks - 1) * 32 + b_krow[j];
      o = o < a.w_cout ? o : a.w_cout - 1;
      b_voff_last[j] = (unsigned)((n + (o - (cchunks - 1) * 32) * a.w_taps * a.w_cin) * 4);
    }
  }
  const bool ragged_cout = DGRAD && (a.w_cout & 31) != 0;
  // DMA instruction idx (A: 0..A_INSTR-1, W: A_INSTR..) goes behind MFMA group idx*3/total: groups 0-2
  // carry everything, group 3 nothing, so the last DMA has a whole group (>= 16 MFMAs) to land before
  // the tile-end vmcnt(0)
  constexpr int DMA_TOTAL = A_INSTR + B_INSTR;
  constexpr int DMA_GROUPS = 3;
  auto stage2 = [&](int buf, int part) {
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j)
      if (part < 0 || (j * DMA_GROUPS) / DMA_TOTAL == part)
        lds_dma16_s(a_voff[j], a_sbase + a_koff, lds0 + buf * STAGE + (j * NW + wave) * 1024);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      if (!(part < 0 || ((A_INSTR + j) * DMA_GROUPS) / DMA_TOTAL == part)) continue;
      // (a select on a scalar condition, not a branch: control flow inside the unrolled body made hipcc copy the
      // 64 accumulator registers between basic blocks)
      const unsigned voff = (DGRAD && ragged_cout && n_lastcc) ? b_voff_last[j] : b_voff[j];
      lds_dma16_s(voff, b_sbase + b_koff, lds0 + buf * STAGE + A_BYTES + (j * NW + wave) * 1024);
    }
  };
  // `parity` (which half of the double buffer holds this tile) is a compile-time constant: the k-loop is unrolled by
  // two, so the buffer offset folds into the ds_read immediates instead of one vector add per fragment read
  // (NS = 4: `parity` is the stage 0..3 holding this k-step, the prefetch goes three k-steps ahead into the stage the
  // previous k-step just released, and `waitn` = DMA instructions that may stay in flight — loads return in order)
  auto ktile = [&](auto parity, auto prefetch, auto waitn) {
    constexpr bool PREFETCH = decltype(prefetch)::value;
    constexpr int PAR = decltype(parity)::value;
    constexpr int WAITN = decltype(waitn)::value;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
    if (NS == 2) __syncthreads();
    else lds_barrier();
    if (PREFETCH) advance();
    constexpr int nbuf = (PAR + NS - 1) % NS;
    const unsigned char* sA = smem + PAR * STAGE;
    const unsigned char* sB = sA + A_BYTES;
    f32x4 af[2][TM], bf[2][TN];
    float bs[2][TN][4];
    auto load_group = [&](int g, int buf) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + lrow;
        af[buf][i] = *reinterpret_cast<const f32x4*>(sA + row * 128 + (((2 * g + h) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!DGRAD) {
          const int row = (wn * TN + j) * 32 + lrow;
          bf[buf][j] = *reinterpret_cast<const f32x4*>(sB + row * 128 + (((2 * g + h) ^ ((row >> 1) & 7)) << 4));
        } else {
          const int col = (wn * TN + j) * 32 + lrow;
          const float* tb = reinterpret_cast<const float*>(sB);
#pragma unroll
          for (int st = 0; st < 4; ++st) bs[buf][j][st] = tb[(g * 8 + 4 * h + st) * BN + col];
        }
      }
    };
    auto mfma_step = [&](int buf, int st) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][i][st], DGRAD ? bs[buf][j][st] : bf[buf][j][st],
                                                           acc[i][j], 0, 0, 0);
    };
    load_group(0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cur = g & 1;
      mfma_step(cur, 0);
      mfma_step(cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (PREFETCH) stage2(nbuf, g);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(cur, 2);
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < 4) load_group(g + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(cur, 3);
    }
  };
  // ---- runs of the pinned summation order (include/vy_math.h vy_conv_k_chunks): every ch_len k-steps (absolute position in
  // the tile's K sequence) the chain in `acc` is complete.  It is PARKED — 16-byte stores into slot c of the
  // tile's scratch (a.ck_scratch: [run c][tile][BM x BN]), fire and forget: no wait, the DMA pipeline keeps running — and
  // `acc` restarts from +0.  Stream-K instances store write-through (sc1) and read back past L1: a chain parked by the
  // block that starts a tile is read by the block that finishes it; plain instances use the default policy.  (Measured:
  // the stores are 0.2 % of the batch-64 step either way; what costs is reading the chains back — see the end of the tile.)  When the tile's last run is done its workgroup reads the parked chains back and adds them in
  // run order, (((+0 + P0) + P1) + P2) + acc.  The lanes that read a slot are the lanes that wrote it — or, for a stream-K
  // tile, lanes of the block that finishes it, after the hand-off flag (the HEAD block drains its stores before raising it).
  // (First build: a running sum R += P_c at every boundary.  The load in it has to return before the next MFMA, and
  // loads return in order — so every boundary drained the LDS-DMA queue: +13 us on a 75 us launch with one block per CU.)
  // Boundaries sit at multiples of NS k-steps from kb (ch_len % 4 == 0; the stream-K schedule cuts chunked launches at
  // multiples of 4).  ch_len == 0: one run (K < 2048, data gradients, and the workgroups of a split-K launch).
  // CK: instances for the launches whose K is summed in runs.  A template parameter, not just ch_len > 0: with the run
  // bookkeeping compiled into every forward instance the batch-64 step lost 0.6 % on launches that have ONE run (more
  // scalar state in the k-loop, 11 more spilled SGPRs in the stream-K instances; same-box A/B, profiles/r06_runs_ab.txt)
  const bool chunked = CK && !DGRAD && ch_len > 0;
  int nb = 0x7fffffff;  // k-step (relative to kb) of the next boundary
  int run = 0;          // the run `acc` is the chain of
  if (chunked) {
    // (an integer division runs on the vector unit: its result is made scalar again by hand, or the loop bounds below —
    // and with them every address of the k-loop — would count as per-lane values)
    run = __builtin_amdgcn_readfirstlane(kb / ch_len);
    const int r = kb - run * ch_len;
    if (r == 0 && kb > 0) --run;  // a piece that starts ON a boundary holds the finished chain of the run before it
    nb = __builtin_amdgcn_readfirstlane((r == 0 && kb > 0) ? 0 : ch_len - r);
  }
  auto ck_slot = [&](int c) {
    return __builtin_amdgcn_make_buffer_rsrc(a.ck_scratch + ((size_t)c * n_tiles + v) * (BM * BN), 0, BM * BN * 4, 0x00020000);
  };
  auto boundary = [&]() {
    const __amdgpu_buffer_rsrc_t wr = ck_slot(run);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 p4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            p4[e] = acc[i][j][q * 4 + e];
            acc[i][j][q * 4 + e] = 0.0f;
          }
          const unsigned off = (unsigned)((((i * TN + j) * 4 + q) * NT + tid) * 16);
#if !(defined(VY_CK_ABL) && (VY_CK_ABL & 2))  // measurement builds: what do the parking stores cost?
          if constexpr (SK) buf_store_f32x4_sc1(p4, wr, off);
          else buf_store_f32x4(p4, wr, off);
#endif
        }
    ++run;
  };
  using W0 = std::integral_constant<int, 0>;
  if constexpr (NS == 2) {
    advance();
    stage2(0, -1);
    VY_TRACE(1)
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 2 < T; t += 2) {
      if (chunked && t == nb) {  // (uniform: a scalar compare and branch per pair of k-steps)
        boundary();
        nb += ch_len;
      }
      ktile(P0{}, std::true_type{}, W0{});
      ktile(P1{}, std::true_type{}, W0{});
    }
    if (chunked && t == nb && t < T) {  // (a boundary exactly where the peeled tail starts)
      boundary();
      nb += ch_len;
    }
    if (t + 2 == T) {
      ktile(P0{}, std::true_type{}, W0{});
      ktile(P1{}, std::false_type{}, W0{});
    } else {
      ktile(P0{}, std::false_type{}, W0{});
    }
  } else {
    static_assert(NS == 2 || NS == 4, "2 or 4 stages");
    using W1 = std::integral_constant<int, DMA_TOTAL>;       // one later k-step may still be in flight
    using W2 = std::integral_constant<int, 2 * DMA_TOTAL>;   // two
    for (int g = 0; g < NS - 1 && g < T; ++g) {  // k-steps 0..2 go out before any matrix work
      advance();
      stage2(g, -1);
    }
    VY_TRACE(1)
    // k-step t: stage t % 4; prefetches k-step t + 3 if there is one; k-steps t+1 and t+2 (if any) stay in flight
    auto kstep = [&](auto stage, int t) {
      const int after = T - 1 - t;
      if (after >= 3) ktile(stage, std::true_type{}, W2{});
      else if (after == 2) ktile(stage, std::false_type{}, W2{});
      else if (after == 1) ktile(stage, std::false_type{}, W1{});
      else ktile(stage, std::false_type{}, W0{});
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    int t = 0;
    for (; t + 7 <= T; t += 4) {  // all four have a k-step to prefetch (t + 3 + 3 < T)
      if (chunked && t == nb) {
        boundary();
        nb += ch_len;
      }
      ktile(S0{}, std::true_type{}, W2{});
      ktile(S1{}, std::true_type{}, W2{});
      ktile(S2{}, std::true_type{}, W2{});
      ktile(S3{}, std::true_type{}, W2{});
    }
    for (; t < T; t += 4) {
      if (chunked && t == nb) {
        boundary();
        nb += ch_len;
      }
      kstep(S0{}, t);
      if (t + 1 < T) kstep(S1{}, t + 1);
      if (t + 2 < T) kstep(S2{}, t + 2);
      if (t + 3 < T) kstep(S3{}, t + 3);
    }
  }

  VY_TRACE(2)
#if defined(VY_CK_ABL) && (VY_CK_ABL & 1)  // measurement builds: what does reading the parked chains back cost?
  if (false) {
#else
  if (chunked && !(SK && store_partial) && run > 0) {
#endif
    // the tile's last run is complete: out = (((+0 + P0) + P1) + P2) + acc, the parked chains read back in run order.
    // Memory-level parallelism is what this costs (all resident blocks of a launch reach it at about the same time, so the
    // neighbour on the CU does not hide it): the three slots of TWO accumulator tiles are requested before anything is
    // added — 24 loads of 16 bytes per lane in flight — through descriptors whose range is 0 for the runs this tile does
    // not have (those loads return +0 without touching memory, and x + (+0) = x for every x a chain from +0 can hold: a
    // chain never yields -0).  One fixed, branch-free sequence for 2 and 4 runs.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own parking stores are in L2 (the DMA queue is empty here anyway)
    constexpr int MAXP = 3;  // vy_conv_k_chunks() - 1 at most
    __amdgpu_buffer_rsrc_t slot[MAXP];
#pragma unroll
    for (int c = 0; c < MAXP; ++c)
      slot[c] = __builtin_amdgcn_make_buffer_rsrc(a.ck_scratch + ((size_t)(c < run ? c : 0) * n_tiles + v) * (BM * BN), 0,
                                                  c < run ? BM * BN * 4 : 0, 0x00020000);
    constexpr int NG = TM * TN, GSTEP = NG >= 2 ? 2 : 1;
#pragma unroll
    for (int g0 = 0; g0 < NG; g0 += GSTEP) {
      f32x4 pk[GSTEP][MAXP][4];
#pragma unroll
      for (int gg = 0; gg < GSTEP; ++gg)
#pragma unroll
        for (int c = 0; c < MAXP; ++c)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned off = (unsigned)((((g0 + gg) * 4 + q) * NT + tid) * 16);
            // stream-K instances park and read back write-through / past L1 throughout (a chain may have been parked by
            // the block that started the tile); plain instances with the default policy (the same lanes read it back)
            if constexpr (SK) pk[gg][c][q] = buf_load_f32x4_sc1(slot[c], off);
            else pk[gg][c][q] = buf_load_f32x4(slot[c], off);
          }
#pragma unroll
      for (int gg = 0; gg < GSTEP; ++gg) {
        const int gi = (g0 + gg) / TN, gj = (g0 + gg) % TN;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float t = 0.0f;
#pragma unroll
            for (int c = 0; c < MAXP; ++c) t = t + pk[gg][c][q][e];
            acc[gi][gj][q * 4 + e] = t + acc[gi][gj][q * 4 + e];
          }
      }
    }
  }
  if constexpr (SK) {
    if (ks_reduce > 0) {
      // split-K: this block ran the LAST chunk.  The blocks of chunks 0 .. ks_reduce - 1 (slab / flag index c * tiles + v)
      // raise their flags without waiting for anything; add their sums in chunk order, then this block's own
      if (tid == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int c = 0; c < ks_reduce; ++c)
          while (__hip_atomic_load(sk.flags + (c * sk.tiles + v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) __builtin_trap();
          }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned off = (unsigned)((((i * TN + j) * 4 + q) * NT + tid) * 16);
            f32x4 t4 = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < ks_reduce; ++c) {
              const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc(
                  sk.partials + (size_t)(c * sk.tiles + v) * (BM * BN), 0, BM * BN * 4, 0x00020000);
              const f32x4 p4 = buf_load_f32x4_sc1(slab, off);
#pragma unroll
              for (int e = 0; e < 4; ++e) t4[e] = t4[e] + p4[e];   // (((+0 + P0) + P1) + ...: the pinned order, vy_math.h)
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][q * 4 + e] = t4[e] + acc[i][j][q * 4 + e];
          }
      __syncthreads();
      if (tid == 0)
        for (int c = 0; c < ks_reduce; ++c)
          __hip_atomic_store(sk.flags + (c * sk.tiles + v), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (SK && load_partial && tid == 0)  // every thread passed a k-loop barrier after loading its partial sums
    __hip_atomic_store(sk.flags + (vblk - 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (SK && store_partial) {
    // head piece: the accumulators go to this block's slab as 16-byte write-through stores (a wave instruction covers
    // 1 KiB of whole lines), every wave drains its own, the barrier, then one lane raises the flag — no release fence
    // (it would write back every dirty line of the XCD's L2, the other blocks' conv outputs included: the first build
    // did, and a hand-off cost 12 us)
    const __amdgpu_buffer_rsrc_t slab_out =
        __builtin_amdgcn_make_buffer_rsrc(sk.partials + (size_t)vblk * (BM * BN), 0, BM * BN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v4;
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] = acc[i][j][q * 4 + e];
          buf_store_f32x4_sc1(v4, slab_out, (unsigned)((((i * TN + j) * 4 + q) * NT + tid) * 16));
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(sk.flags + vblk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
#if defined(VY_CONV_ABLATE) && (VY_CONV_ABLATE & 1)  // tools/probe/conv_tile_trace.hip only: what would a free epilogue buy?
  {
    float s_ = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_ += acc[i][j][r];
    if (s_ == 12345.678f) a.out[tid] = s_;
    VY_TRACE(3)
    return;
  }
#endif
  // epilogue: affine (folded BN or bias) -> leaky -> + addend -> store (x1 or x2-replicated), through
  // buffer descriptors based at the tile's first pixel and column (see the row tables above)
  constexpr int kRsrcFlags = 0x00020000;  // raw dword buffer, gfx9 data format 32
  const __amdgpu_buffer_rsrc_t out_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(a.out + (pix0 * a.o_cs + a.o_co + n0), 0, 0x7fffffff, kRsrcFlags);
  const __amdgpu_buffer_rsrc_t res_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res ? a.res + (pix0 * a.r_cs + a.r_co + n0) : a.in), 0, 0x7fffffff, kRsrcFlags);
  const int ups_dx = a.o_cs * 4, ups_dy = a.o_Wp * a.o_cs * 4;
  // The epilogue variant (which of scale / shift / leaky / addend / x2-replicate apply) is uniform for the
  // launch; it is resolved ONCE, outside the element loops, into a straight-line instance per combination —
  // with the tests inside the loops hipcc emitted a branch (and a serialising wait) per element.
  auto epilogue = [&](auto has_scale_, auto has_shift_, auto leaky_, auto has_res_, auto ups2_) {
    constexpr bool has_scale = decltype(has_scale_)::value, has_shift = decltype(has_shift_)::value;
    constexpr bool leaky = decltype(leaky_)::value, has_res = decltype(has_res_)::value, ups2 = decltype(ups2_)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ncol = (wn * TN + j) * 32 + lrow;
      const int n = n0 + ncol;
      const bool nvalid = n < a.N;
      const int nc = nvalid ? n : a.N - 1;
      // the lane's column offset with bit 31 set for a column past N; added to a row offset with a SATURATING add, so
      // that "invalid row" + "invalid column" stays out of the descriptor's range (one instruction per address)
      const unsigned colc = (unsigned)ncol * 4u | (nvalid ? 0u : kInvalidRow);
      float sc = 1.0f, sh = 0.0f;
      if (has_scale) sc = a.scale[nc];
      if (has_scale || has_shift) sh = a.shift[nc];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        unsigned oo[16];
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          oo[r] = __builtin_elementwise_add_sat(o_off[row], colc);
          if (has_res) rv[r] = buf_load_f32(res_rsrc, __builtin_elementwise_add_sat(r_off[row], colc));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float vv = acc[i][j][r];
          if (has_scale)
            vv = fmaf(vv, sc, sh);
          else if (has_shift)
            vv = vv + sh;
          if (leaky) vv = vy_leaky(vv);
          if (has_res) vv = vv + rv[r];
          if (!ups2) {
            buf_store_f32(vv, out_rsrc, oo[r], 0);
          } else {  // an offset with bit 31 set is out of the descriptor's range: that store is dropped
            const unsigned base = oo[r] & ~3u;
            const unsigned no_dx = (oo[r] & 1u) << 31, no_dy = (oo[r] & 2u) << 30;
            buf_store_f32(vv, out_rsrc, base, 0);
            buf_store_f32(vv, out_rsrc, base | no_dx, ups_dx);
            buf_store_f32(vv, out_rsrc, base | no_dy, ups_dy);
            buf_store_f32(vv, out_rsrc, base | no_dx | no_dy, ups_dy + ups_dx);
          }
        }
      }
    }
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    const bool f_scale = a.scale != nullptr, f_shift = a.shift != nullptr, f_leaky = a.leaky != 0;
    const bool f_res = a.res != nullptr, f_ups2 = a.ups == 2;
    if (f_scale && f_shift && f_leaky && !f_ups2) {           // conv + BN + leaky (+ residual): inference cells
      if (f_res) epilogue(T{}, T{}, T{}, T{}, F{});
      else epilogue(T{}, T{}, T{}, F{}, F{});
    } else if (f_scale && f_shift && f_leaky && !f_res) {     // transition cells: x2-replicated store
      epilogue(T{}, T{}, T{}, F{}, T{});
    } else if (!f_scale && !f_leaky && !f_ups2) {             // raw conv / bias / gradients (+ accumulate)
      if (f_shift) {
        if (f_res) epilogue(F{}, T{}, F{}, T{}, F{});
        else epilogue(F{}, T{}, F{}, F{}, F{});
      } else {
        if (f_res) epilogue(F{}, F{}, F{}, T{}, F{});
        else epilogue(F{}, F{}, F{}, F{}, F{});
      }
    } else {
      __builtin_trap();  // no launch site builds any other combination (vy_launch_conv_igemm rejects them)
    }
  }

  VY_TRACE(3)
#ifdef VY_CONV_TRACE
  if (a.trace && tid == 0) {
    a.trace[(long long)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg(4 | (31 << 11));   // HW_ID
    a.trace[(long long)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg(20 | (31 << 11));  // XCC_ID
  }
#endif
  // train-mode BatchNorm: per-tile column sums of the raw accumulators (deterministic: fixed
  // order inside the tile, tiles are combined in order by the finalize kernel)
  if (a.stats) {
    // double accumulation: var = E[x^2] - mean^2 cancels badly in fp32 when |mean| >> std, and the
    // batch statistics must round to the same fp32 mean / var as the CPU checker's
    double s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] = 0.0;
      s2[j] = 0.0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          const double vv = o_off[row] != kInvalidRow ? (double)acc[i][j][r] : 0.0;
          s1[j] += vv;
          s2[j] += vv * vv;
        }
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
    }
    __syncthreads();  // every wave is past its last LDS tile read
    double* red = reinterpret_cast<double*>(smem);  // [WM][2][BN]
    if (h == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + lrow;
        red[(wm * 2 + 0) * BN + col] = s1[j];
        red[(wm * 2 + 1) * BN + col] = s2[j];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.N) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        t1 += red[(w * 2 + 0) * BN + tid];
        t2 += red[(w * 2 + 1) * BN + tid];
      }
      a.stats[((long long)tile_m * 2 + 0) * a.N + n0 + tid] = t1;
      a.stats[((long long)tile_m * 2 + 1) * a.N + n0 + tid] = t2;
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

template <int BM, int BN, int WM, int WN, bool DGRAD, int NS = 2, bool SK = false, bool CK = false>
// SK = chain-preserving stream-K.  A plain launch hands whole tiles to the CUs, so a launch of 2.66 x 256 tiles costs three
// rounds.  The SK instance is launched with exactly as many blocks as the chip holds; the sequence of all k-steps (tile
// 0's, tile 1's, ...) is cut into equal contiguous shares, one per block, so a block owns [the tail of a tile][whole
// tiles][the head of a tile].  It runs them in the order HEAD, whole tiles, TAIL: the head piece (k-steps 0 .. k1 of its
// last tile) first — its 16 accumulator registers per MFMA tile go to a scratch slab and a flag is raised; the tail piece
// (k-steps k0 .. T of its first tile) last — it waits for the previous block's flag (raised long ago: that block did the
// head first), loads the accumulators as the MFMA C operand and continues the SAME fma chain, then runs the normal
// epilogue.  Every output element is still one fma chain in the pinned order: results are bit-identical to the plain launch.
// NS = LDS stages.  2: the throughput configuration (2-4 resident blocks per CU hide each other's DMA latency).
// 4: for launches of so few blocks that a CU holds one (batch 1, the 13x13 maps at batch 16): the k-step of a lone
// 64x64 block is 0.43 us of matrix work but a DMA takes ~0.8 us to land, so three k-steps are kept in flight.
// 2nd launch-bounds argument = waves per SIMD the register allocation must allow: two (2 blocks/CU of 4
// waves, or one 8-wave block).  Without it hipcc let the register count drift past 256 and silently
// halved the occupancy of some variants.
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_igemm_kernel(const ConvArgs a, const int tiles_n, const SkArgs sk) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub (amdgcn builtins below)
  constexpr int STAGE = (BM + BN) * 128;
  // one LDS object: [stage0 A|W][stage1 A|W]...[row tables]
#ifndef VY_CONV_PAD_LDS  // probe builds only (tools/probe): extra LDS per block to lower the blocks a CU holds
#define VY_CONV_PAD_LDS 0
#endif
  __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE + 2 * BM * 8 + (NS == 2 ? VY_CONV_PAD_LDS : 0)];
  // XCD-aware order: blocks L, L+8, ... share an XCD (L2); give each XCD a contiguous run of tiles (SK: of the
  // k-step sequence) with n fastest so neighbours in time re-use the same A rows and the whole W panel.
  int vblk;
  {
    const int nblk = gridDim.x, L = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = L & 7, idx = L >> 3;
    vblk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int T_all = a.ntaps * (a.Kc >> 5);  // k-steps of a whole tile


  if constexpr (!SK) {
    conv_tile<BM, BN, WM, WN, DGRAD, NS, SK, CK, const ConvArgs>(a, tiles_n, sk, smem, vblk, vblk, 0, T_all, false, false, 0,
                                                                 CK ? a.k_chunk : 0, (int)gridDim.x);
  } else {
    // which k-steps of which tiles: sk_schedule.h (shared with the host-side checker of the CPU test suite).
    // (32-bit arithmetic made scalar again by hand: tiles x blocks < 2^31 is checked by the launcher; a 64-bit division
    // is expanded into vector code with control flow and its results then count as per-lane values)
    auto sdiv = [](unsigned n, unsigned d) { return (int)__builtin_amdgcn_readfirstlane((int)(n / d)); };
    if (!CK && sk.ksplit > 0) {
      typedef const __attribute__((address_space(4))) ConvArgs KArgs;
      KArgs* ap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
      const int chunk = sdiv((unsigned)vblk, (unsigned)sk.tiles), tile = vblk - chunk * sk.tiles;
      const int kb = sdiv((unsigned)(chunk * T_all), (unsigned)sk.ksplit), ke = sdiv((unsigned)((chunk + 1) * T_all), (unsigned)sk.ksplit);
      const bool last = chunk == sk.ksplit - 1;
      // (the slab / flag of chunk c's block is index c * tiles + tile == its vblk)
      conv_tile<BM, BN, WM, WN, DGRAD, NS, SK, false, KArgs>(*ap, tiles_n, sk, smem, vblk, tile, kb, ke, false, !last,
                                                       last ? sk.ksplit - 1 : 0);
      return;
    }
    const SkSchedule sch = sk_schedule((int)gridDim.x, (int)blockIdx.x, sk.tiles, T_all, sdiv, (CK && a.k_chunk > 0) ? NS : 1);
    for (int it = 0; it < sch.n_items; ++it) {  // ONE call site: the tile body is instantiated once
      const SkItem w = sk_item(sch, it, T_all);
      const int tile = w.tile, kb = w.kb, ke = w.ke;
      const bool ld = w.load_partial, st = w.store_partial;
      // The arguments are read through the kernarg pointer, made opaque once per item: with `a` itself every field
      // (and everything derived from it) stayed live across the whole loop, the SGPRs ran out and spilled into
      // vector registers (78 -> 151 VGPRs for the 64x64 tile, scratch for 128x128, an occupancy step lost)
      typedef const __attribute__((address_space(4))) ConvArgs KArgs;
      KArgs* ap = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();  // ConvArgs is the first argument
      asm volatile("" : "+s"(ap));
      conv_tile<BM, BN, WM, WN, DGRAD, NS, SK, CK, KArgs>(*ap, tiles_n, sk, smem, vblk, tile, kb, ke, ld, st, 0, CK ? ap->k_chunk : 0,
                                                          sk.tiles);
    }
  }
#endif  // __HIP_DEVICE_COMPILE__
}

// blocks of an instance one CU holds, and the CU count (queried once)
template <typename K>
static int resident_blocks(K kernel, int threads) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess || n < 1) n = 1;
  return n;
}
// CUs of the current device (per device, queried once each; VY_CU_COUNT=n overrides it: tests of the cost models'
// behaviour on a chip they were not fitted on).  256 when there is no device to ask (host-only callers)
int vy_cu_count_of(int dev) {
  static const int forced = getenv("VY_CU_COUNT") ? atoi(getenv("VY_CU_COUNT")) : 0;
  if (forced > 0) return forced;
  static std::atomic<int> cache[64];  // (several host threads may drive their own nets: relaxed atomics, idempotent fill)
  if (dev < 0 || dev >= 64) return VY_MODEL_CUS;
  int c = cache[dev].load(std::memory_order_relaxed);
  if (!c) {
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c < 1) c = VY_MODEL_CUS;
    cache[dev].store(c, std::memory_order_relaxed);
  }
  return c;
}
int vy_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return vy_cu_count_of(-1);
  return vy_cu_count_of(dev);
}
int vy_cu_count_of_ptr(const void* p) {
  hipPointerAttribute_t at;
  if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (at.type != hipMemoryTypeDevice) return 0;
  return vy_cu_count_of(at.device);
}

// ---- stream-K topology check --------------------------------------------------------------------------------------
// The stream-K schedule (sk_schedule.h) is built for the MI355X in SPX mode: 256 CUs, 8 XCDs, workgroups dealt round-robin
// over the XCDs so that blocks L and L + 8 share an L2 (its locality, its cost model and its "one XCD group = one run of
// tiles" bookkeeping all assume that).  The hand-off itself is placement-independent (write-through sc1 stores, sc1 loads:
// MI355X_MICROARCH.md, Valid forms), but nothing else about the schedule has been measured on another partition mode, a
// CU-masked queue or a part with fewer XCDs — so a net enables stream-K only after this probe has seen the expected
// placement on its device: 512 one-wave blocks record their XCC id; ids must be < 8, the first eight distinct, and
// id[b] == id[b % 8] for every b.  Otherwise the net runs plain launches (the SK scratch pointers stay null).
__global__ void xcc_probe_kernel(unsigned* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15u;  // HW_REG_XCC_ID
#endif
}

int vy_sk_verify_topology(unsigned* scratch_dev, hipStream_t s) {
  static int cache[64];  // per device: 0 unknown, 1 verified, 2 refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
  if (cache[dev]) return cache[dev] == 1;
  if (getenv("VY_CONV_SK_NO_TOPOLOGY_CHECK") && atoi(getenv("VY_CONV_SK_NO_TOPOLOGY_CHECK"))) return (cache[dev] = 1) == 1;
  constexpr int NB = 512;
  unsigned host[NB];
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return 0;  // decided outside a capture
  hipLaunchKernelGGL(xcc_probe_kernel, dim3(NB), dim3(64), 0, s, scratch_dev);
  bool ok = hipGetLastError() == hipSuccess &&
            hipMemcpyAsync(host, scratch_dev, sizeof host, hipMemcpyDeviceToHost, s) == hipSuccess &&
            hipMemsetAsync(scratch_dev, 0, sizeof host, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
  ok = ok && vy_cu_count_of(dev) == 256;  // the cost model's rounds and the schedule's shares are per 256 CUs
  for (int b = 0; ok && b < NB; ++b) ok = host[b] < 8u && host[b] == host[b & 7];
  for (int i = 0; ok && i < 8; ++i)
    for (int j = 0; j < i; ++j) ok = ok && host[i] != host[j];
  cache[dev] = ok ? 1 : 2;
  return ok;
}

// ---- the launch cost model (conv_cost_model.h), shared by the tile choice (select_cfg) and the stream-K decision ----
struct SkSwitches {
  int on, dgrad, slots;
  double min_gain, cost;
};
// VY_CONV_SK=0: plain launches only.  VY_CONV_SK_DGRAD=1: data gradients too (off: in the training step the weight-
// gradient stream runs beside them and its blocks already fill the CUs a partly filled round leaves idle; measured,
// stream-K there only moves the idle time).  VY_CONV_SK_SLOTS=n (tests): EVERY launch of more than n tiles on n blocks.
// VY_CONV_SK_GAIN / VY_CONV_SK_COST: the criterion's threshold (3 % of the plain launch) and hand-off cost (7.5 us).
static const SkSwitches& sk_switches() {
  static const SkSwitches s = {getenv("VY_CONV_SK") ? atoi(getenv("VY_CONV_SK")) : 1,
                               getenv("VY_CONV_SK_DGRAD") ? atoi(getenv("VY_CONV_SK_DGRAD")) : 0,
                               getenv("VY_CONV_SK_SLOTS") ? atoi(getenv("VY_CONV_SK_SLOTS")) : 0,
                               getenv("VY_CONV_SK_GAIN") ? atof(getenv("VY_CONV_SK_GAIN")) : 0.03,
                               getenv("VY_CONV_SK_COST") ? atof(getenv("VY_CONV_SK_COST")) : 7.5};
  return s;
}
static bool sk_allowed(const ConvArgs& a) {
  const SkSwitches& w = sk_switches();
  return w.on && a.sk_partials && a.sk_flags && (!a.dgrad || w.dgrad || w.slots > 0);
}
static VySkPolicy sk_policy(const ConvArgs& a) {
  return VySkPolicy{sk_allowed(a), sk_switches().min_gain, sk_switches().cost};
}

// `sk_query` != nullptr: nothing is launched, *sk_query tells whether the launch would be a stream-K one (the profile's label)
template <int BM, int BN, int WM, int WN, int NS = 2>
static hipError_t launch_cfg(const ConvArgs& a_in, hipStream_t s, bool* sk_query = nullptr, int* ks_query = nullptr) {
  ConvArgs a = a_in;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
  const long long tiles = (long long)tiles_m * tiles_n;
  SkArgs sk = {nullptr, nullptr, 0, 0};
  // the runs of the pinned summation order (forward launches only; gradients are not held to the oracle's bits)
  const int T_runs = a.ntaps * (a.Kc >> 5);
  const int ksplit_S = a.dgrad ? 1 : vy_conv_runs(a.ntaps, a.Kc >> 5);
  a.k_chunk = 0;
  if (ks_query) *ks_query = 0;
  if (ksplit_S > 1) {
    a.k_chunk = T_runs / ksplit_S;
    // a workgroup that runs more than one run of a tile parks the finished runs' sum in the tile's scratch
    if (!sk_query && (!a.ck_scratch || (ksplit_S - 1) * tiles * BM * BN * 4ll > (long long)a.ck_bytes)) return hipErrorInvalidValue;
  }
  // Stream-K when the cost model says it pays (predict_launch): the launches whose last round of the CUs is poorly
  // filled.  ON by default for forward launches since the hand-off is write-through and the schedule keeps a plain
  // launch's locality (profiles/r03_negative_results.txt section 2 has the two builds that lost and why): 608x608
  // batch 64 +0.7 %, 416x416 +0.6 %, the training forward -0.3 ms; VY_CONV_SK=0 restores plain launches.
  if (sk_allowed(a)) {
    static const int res_f = resident_blocks(conv_igemm_kernel<BM, BN, WM, WN, false, NS, true>, WM * WN * 64);
    static const int res_d = resident_blocks(conv_igemm_kernel<BM, BN, WM, WN, true, NS, true>, WM * WN * 64);
    const int sk_slots = sk_switches().slots;
    const int cus = vy_args_cus(a);
    // blocks per CU: as many as fit, but never so many that a share is shorter than one tile (a share is at most
    // [head][whole tiles][tail]).  344 tiles (the 13x13 maps at batch 16): ONE block per CU with 1.34 tiles each, where
    // the plain launch leaves 88 CUs with two tiles and 168 with one
    const long long per_cu = std::min<long long>(a.dgrad ? res_d : res_f, tiles / cus);
    const long long G = sk_slots > 0 ? sk_slots : (long long)cus * per_cu;
    // Split-K: a launch whose K is summed in S > 1 runs (the pinned order, vy_math.h) and whose tiles leave at least half
    // of the CUs empty goes out as tiles x S workgroups, one run each; the last run's workgroup adds the others' sums in
    // run order (conv_tile, ks_reduce) — the same arithmetic as one workgroup running the runs one after the other.
    // Measured (profiles/r06_ksplit_probe.txt, one frame): 13x13 / 19x19 3x3 cells on 512 channels 75 -> 33 / 47 us; a
    // launch of 184 tiles cut in two lost 4 %: hence tiles <= CUs / 2.  VY_CONV_KSPLIT=0: never.
    if (ksplit_S > 1) {
      static const int ks_on = getenv("VY_CONV_KSPLIT") ? atoi(getenv("VY_CONV_KSPLIT")) : 1;
      const long long cap = (long long)cus * res_f;
      if (ks_on && sk_slots == 0 && 2 * tiles <= cus && tiles * ksplit_S <= cap && tiles * ksplit_S * BM * BN * 4ll <= (long long)a.sk_bytes &&
          tiles * ksplit_S <= a.sk_nflags) {
        if (sk_query) {
          *sk_query = false;
          if (ks_query) *ks_query = ksplit_S;
          return hipSuccess;
        }
        sk.partials = a.sk_partials;
        sk.flags = a.sk_flags;
        sk.tiles = (int)tiles;
        sk.ksplit = ksplit_S;
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, NS, true>), dim3((unsigned)(tiles * ksplit_S)),
                           dim3(WM * WN * 64), 0, s, a, tiles_n, sk);
        return hipGetLastError();
      }
    }
    bool pays = false;
    if (const VyTileModel* tm = vy_tile_model(BM, BN))
      vy_predict_launch(a.M, a.N, (double)a.ntaps * a.Kc, *tm, sk_policy(a), &pays, cus);
    if (G > 0 && tiles > G && tiles * (G + 8) < (1ll << 31) && 2 * (G / 8 + 1) * (long long)(a.ntaps * (a.Kc >> 5)) < (1ll << 31) &&
        (sk_slots > 0 || pays) && G * BM * BN * 4ll <= (long long)a.sk_bytes && G <= a.sk_nflags) {
      if (sk_query) {
        *sk_query = true;
        return hipSuccess;
      }
      sk.partials = a.sk_partials;
      sk.flags = a.sk_flags;
      sk.tiles = (int)tiles;
      if (a.dgrad)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true, NS, true>), dim3((unsigned)G), dim3(WM * WN * 64), 0, s,
                           a, tiles_n, sk);
      else if (a.k_chunk > 0)
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, NS, true, true>), dim3((unsigned)G), dim3(WM * WN * 64), 0, s,
                           a, tiles_n, sk);
      else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, NS, true>), dim3((unsigned)G), dim3(WM * WN * 64), 0, s,
                           a, tiles_n, sk);
      return hipGetLastError();
    }
  }
  if (sk_query) {
    *sk_query = false;
    return hipSuccess;
  }
  if (a.dgrad)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true, NS>), dim3(tiles_m * tiles_n), dim3(WM * WN * 64), 0, s,
                       a, tiles_n, sk);
  else if (a.k_chunk > 0)
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, NS, false, true>), dim3(tiles_m * tiles_n), dim3(WM * WN * 64), 0,
                       s, a, tiles_n, sk);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false, NS>), dim3(tiles_m * tiles_n), dim3(WM * WN * 64), 0, s,
                       a, tiles_n, sk);
  return hipGetLastError();
}

// Tile choice: conv_cost_model.h, plus the experiment switches.
static void select_cfg(const ConvArgs& a, int* bm, int* bn) {
  auto blocks = [&](int m, int n) { return (long long)((a.M + m - 1) / m) * ((a.N + n - 1) / n); };
  // experiment switch (tools/train_layers.sh): VY_CONV_FORCE=128x64 runs every launch on that tile
  static const char* force = getenv("VY_CONV_FORCE");
  if (force && sscanf(force, "%dx%d", bm, bn) == 2 && !(*bm == 32 && a.dgrad)) return;
  if (a.N <= 32) {
    *bm = 128;
    *bn = 32;
    return;
  }
  const double K = (double)a.ntaps * a.Kc;
  bool sk_unused;
  double best = vy_select_tile(a.M, a.N, K, sk_policy(a), bm, bn, &sk_unused, vy_args_cus(a));
  // 16x16 wave tiles (conv_small.hip; block tile 32 x {32, 64}): OFF by default.  Measured on the MI355X (round 3,
  // profiles/r03_negative_results.txt): bit-exact, but 1.9 - 2.0x SLOWER than the 64x64 tile on the batch-1 3x3 layers it
  // was built for (76x76: 78-85 vs 42 us, 38x38: 85-88 vs 43, 19x19: 98-128 vs 77) — a 32-channel sub-step of a 32x32
  // block is 8 MFMAs per wave against 2 LDS-DMA instructions (~95 cycles of issue each), 8 fragment reads, 16 selects
  // and a barrier; the matrix pipe waits for the instruction stream, not the other way round.  VY_CONV_SMALL=1 enables
  // the choice by this model (kept for experiments; VY_CONV_FORCE=32x32 / 32x64 forces the kernel for parity runs).
  static const int small_on = getenv("VY_CONV_SMALL") ? atoi(getenv("VY_CONV_SMALL")) : 0;
  if (small_on && !a.dgrad && blocks(64, 64) <= 2048) {
    const int bns[2] = {32, 64};
    for (int bn_s : bns) {
      if (bn_s == 64 && a.N <= 32) continue;
      const long long nb = blocks(32, bn_s);
      const long long per_cu = (nb + 255) / 256;
      const double units = bn_s / 32.0;  // 16x16 accumulators per wave
      double t = (per_cu * units > 1.0 ? per_cu * units * 0.00333 : 0.00417) * K + 2.5 * (double)((per_cu + 1) / 2);
      if (t < best * 0.97) {
        best = t;
        *bm = 32;
        *bn = bn_s;
      }
    }
  }
}

void vy_conv_cfg(const ConvArgs& a, int* bm, int* bn) { select_cfg(a, bm, bn); }

// predicted time (microseconds) of the launch on this (exact fp32) kernel: what the split-fp32 instance has to beat
double vy_conv_predict_us(const ConvArgs& a) {
  if (a.N <= 32) return 0.0;  // own tile, no model: never handed to the split kernel (cout % 64 != 0 anyway)
  int bm, bn;
  bool sk;
  return vy_select_tile(a.M, a.N, (double)a.ntaps * a.Kc, sk_policy(a), &bm, &bn, &sk, vy_args_cus(a));
}

int vy_conv_tiles_m(const ConvArgs& a) {
  int bm, bn;
  select_cfg(a, &bm, &bn);
  return (a.M + bm - 1) / bm;
}

static VyFastDiv make_fastdiv(unsigned d) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  VyFastDiv f;
  f.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 0 ? l - 1 : 0;
  return f;
}

// tile choice -> template instance; `sk_query`: see launch_cfg
static hipError_t run_cfg(const ConvArgs& a, hipStream_t s, bool* sk_query, int* ks_query = nullptr) {
  int bm, bn;
  select_cfg(a, &bm, &bn);
  if (sk_query) *sk_query = false;
  if (ks_query) *ks_query = 0;
  // (the 16x16-wave-tile kernel computes ONE chain: launches whose K is summed in runs stay on the 32x32 tiles)
  const bool runs = !a.dgrad && vy_conv_runs(a.ntaps, a.Kc >> 5) > 1;
  if (bm == 32 && runs) bm = bn = 64;
  if (bm == 32) return sk_query ? hipSuccess : vy_launch_conv_s16(a, bm, bn, s);
  // experiment switch: VY_CONV_S16=1 sends a forced tile (VY_CONV_FORCE) through the 16x16x4 kernel where it has the instance
  static const int s16_forced = getenv("VY_CONV_S16") ? atoi(getenv("VY_CONV_S16")) : 0;
  if (s16_forced && !runs && !sk_query && !a.dgrad && (bm % 32 == 0) && (bn == 64 || bn == 96)) {
    const hipError_t e = vy_launch_conv_s16(a, bm, bn, s);
    if (e != hipErrorInvalidValue) return e;
  }
  if (bn == 32) return launch_cfg<128, 32, 4, 1>(a, s, sk_query, ks_query);
  if (bm == 128 && bn == 64) return launch_cfg<128, 64, 2, 2>(a, s, sk_query, ks_query);
  if (bm == 64) {
    // few blocks (at most two per CU) and a k-loop long enough to fill it: the four-stage pipeline
    static const int deep = getenv("VY_CONV_DEEP") ? atoi(getenv("VY_CONV_DEEP")) : 512;
    const long long nb = (long long)((a.M + 63) / 64) * ((a.N + 63) / 64);
    if (nb <= deep && a.ntaps * (a.Kc >> 5) >= 8) return launch_cfg<64, 64, 2, 2, 4>(a, s, sk_query, ks_query);
    return launch_cfg<64, 64, 2, 2>(a, s, sk_query, ks_query);
  }
  return launch_cfg<128, 128, 2, 2>(a, s, sk_query, ks_query);
}

bool vy_conv_streamk(const ConvArgs& a) {
  bool sk = false;
  return run_cfg(a, nullptr, &sk) == hipSuccess && sk;
}

int vy_conv_ksplit(const ConvArgs& a) {
  bool sk = false;
  int ks = 0;
  return run_cfg(a, nullptr, &sk, &ks) == hipSuccess ? ks : 0;
}

size_t vy_conv_chunk_scratch_bytes(long long M, int N, int runs) {
  const long long m = (M + 127) / 128 * 128, n = ((long long)N + 127) / 128 * 128;
  return runs > 1 ? (size_t)((runs - 1) * m * n * 4) : 0;
}

hipError_t vy_launch_conv_igemm(const ConvArgs& a_in, hipStream_t s) {
  ConvArgs a = a_in;
  if (a.LW < 1 || a.LH < 1) return hipErrorInvalidValue;
  a.fd_lw = make_fastdiv((unsigned)a.LW);
  a.fd_lh = make_fastdiv((unsigned)a.LH);
  a.pk_dy = a.pk_dx = 0;
  a.pk_w = 0;
  for (int t = 0; t < a.ntaps; ++t) {
    if (a.tap_dy[t] < -1 || a.tap_dy[t] > 1 || a.tap_dx[t] < -1 || a.tap_dx[t] > 1 || a.tap_w[t] > 15) return hipErrorInvalidValue;
    a.pk_dy |= (unsigned)(a.tap_dy[t] + 1) << (2 * t);
    a.pk_dx |= (unsigned)(a.tap_dx[t] + 1) << (2 * t);
    a.pk_w |= (unsigned long long)a.tap_w[t] << (4 * t);
  }
  if (a.Kc % 32 != 0 || a.ntaps < 1 || a.ntaps > 9 || a.M <= 0 || a.N <= 0) return hipErrorInvalidValue;
  if (a.dgrad && (a.N % 4 != 0)) return hipErrorInvalidValue;
  {  // epilogue combinations the kernel instantiates (see `epilogue` in conv_igemm_kernel)
    const bool bn_cell = a.scale && a.shift && a.leaky, plain = !a.scale && !a.leaky;
    if (!((bn_cell && (a.ups != 2 || !a.res)) || (plain && a.ups != 2))) return hipErrorInvalidValue;
  }
  return run_cfg(a, s, nullptr);
}
