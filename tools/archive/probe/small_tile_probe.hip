// What does a 256-output wave tile cost on the fp32 matrix core?  (Batch-1 / small-batch convs: a 19x19 map has 361
// 32x32 wave tiles for 1024 SIMDs; quarter-size tiles are the only way to give every SIMD work.)  Two candidates that
// keep the pinned summation order (channels 0,4,1,5,2,6,3,7 per group of 8; one fma chain per output):
//   A  v_mfma_f32_16x16x4_f32: lane quarter q = lane/16 is k slot q; MFMA #1 of a group takes channels (0,4,1,5), #2
//      (2,6,3,7), so a lane needs channels (c, c+2) of its row: one ds_read_b128 + two v_cndmask per fragment
//   B  v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4x4, k = 1, every lane at the same k: a lane streams its row's channels
//      in chain order straight from two ds_read_b128 (no selects), 8 MFMAs of 8 cycles per group
// Measured here, per SIMD cycle (s_memtime) and per instruction, with 1 and 2 waves per SIMD on every CU:
//   bare dependent chains, and the same chains fed from LDS the way a conv k-loop would feed them.
// Also checks that 4x4x1 is bit-equal to fmaf(a, b, c).
//   hipcc --offload-arch=gfx950 -O3 -o small_tile_probe small_tile_probe.hip && ./small_tile_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 2048;  // groups of 8 channels per wave

template <int MODE>
// 0: 16x16x4 bare chain   1: 16x16x4 + b128 reads + selects   2: 4x4x1 bare chain   3: 4x4x1 + b128 reads
// 4: 16x16x4 + b128 reads, no selects (wrong numbers; isolates the select cost)
__global__ void bench(float* out, long long* cyc, const float* seed) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 32 * 2];  // A tile: 64 rows x 32 floats, B tile behind it
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 32 * 2; i += blockDim.x) lds[i] = seed[i & 1023];
  __syncthreads();
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int q = lane >> 4;
  const bool hi = (q >> 1) != 0;
  const int row16 = (wave & 3) * 16 + (lane & 15);
  const int row4 = (wave & 3) * 16 + ((lane >> 4) & 3) * 4 + (lane & 3);   // 4x4 blocks: A row of block (br = lane/16)
  const int col4 = (wave & 3) * 16 + ((lane >> 2) & 3) * 4 + (lane & 3);   // B row of block column bc
  float a0 = seed[lane], b0 = seed[64 + lane], a1 = seed[128 + lane], b1 = seed[192 + lane];
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for (int it = 0; it < ITERS; ++it) {
    const int g = it & 3;
    if (MODE == 0) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
    } else if (MODE == 1 || MODE == 4) {
      const int chunk = 2 * g + (q & 1);
      const f32x4 av = *reinterpret_cast<const f32x4*>(&lds[row16 * 32 + ((chunk ^ ((row16 >> 1) & 7)) << 2)]);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(&lds[2048 + row16 * 32 + ((chunk ^ ((row16 >> 1) & 7)) << 2)]);
      float x0, x1, y0, y1;
      if (MODE == 1) {
        x0 = hi ? av[1] : av[0];
        x1 = hi ? av[3] : av[2];
        y0 = hi ? bv[1] : bv[0];
        y1 = hi ? bv[3] : bv[2];
      } else {
        x0 = av[0];
        x1 = av[2];
        y0 = bv[0];
        y1 = bv[2];
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, y0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x1, y1, acc, 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b1, acc, 0, 0, 0);
      }
    } else {
      const int sw_a = (row4 >> 1) & 7, sw_b = (col4 >> 1) & 7;
      const f32x4 a_lo = *reinterpret_cast<const f32x4*>(&lds[row4 * 32 + (((2 * g) ^ sw_a) << 2)]);
      const f32x4 a_hi = *reinterpret_cast<const f32x4*>(&lds[row4 * 32 + (((2 * g + 1) ^ sw_a) << 2)]);
      const f32x4 b_lo = *reinterpret_cast<const f32x4*>(&lds[2048 + col4 * 32 + (((2 * g) ^ sw_b) << 2)]);
      const f32x4 b_hi = *reinterpret_cast<const f32x4*>(&lds[2048 + col4 * 32 + (((2 * g + 1) ^ sw_b) << 2)]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // chain order 0,4,1,5,2,6,3,7
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a_lo[k], b_lo[k], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a_hi[k], b_hi[k], acc, 0, 0, 0);
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

__global__ void exact44(const float* A, const float* B, const float* C, float* D) {
  const int l = threadIdx.x;
  f32x4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[l * 4 + r];
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

template <int MODE>
static void run(const char* name, int threads, float* out, long long* cyc, const float* seed, int mfma_per_group, double flop_per_mfma) {
  const int blocks = 256;  // one block per CU (the launch is small enough that placement is one per CU)
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, seed);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, seed);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0;
  for (long long v : h) avg += (double)v;
  avg /= h.size();
  const double n_mfma = (double)ITERS * mfma_per_group;
  const int wps = threads / 256;
  // s_memtime ticks at 100 MHz on gfx9 (constant clock): convert through the launch wall time instead
  const double flops = (double)blocks * (threads / 64) * n_mfma * flop_per_mfma;
  printf("%-44s %d wave/SIMD  %8.3f ms  %7.1f TFLOP/s  (%.2f us per 8-channel group per SIMD)\n", name, wps, ms,
         flops / (ms * 1e-3) / 1e12, ms * 1e3 / ITERS);
}

static float rnd() {
  double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
  return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
}

int main() {
  float *out, *seed;
  long long* cyc;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  std::vector<float> hs(1024);
  srand(3);
  for (float& x : hs) x = rnd() * 0.01f;
  hipMalloc(&seed, 4096);
  hipMemcpy(seed, hs.data(), 4096, hipMemcpyHostToDevice);
  // exactness of 4x4x1: lane l = block l/4, A row i = l%4 / B col j = l%4; D[l][r] = row r, col l%4 of block l/4
  {
    float hA[64], hB[64], hC[256], hD[256], *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    int ok = 0, okT = 0, total = 0;
    for (int trial = 0; trial < 100; ++trial) {
      for (float& x : hA) x = rnd();
      for (float& x : hB) x = rnd();
      for (float& x : hC) x = rnd();
      hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
      hipMemcpy(dC, hC, 1024, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(exact44, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
      hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
          const int blk = l / 4, j = l % 4;
          const float f = fmaf(hA[blk * 4 + r], hB[blk * 4 + j], hC[l * 4 + r]);   // D[row r][col j] of block blk
          const float t = fmaf(hA[blk * 4 + j], hB[blk * 4 + r], hC[l * 4 + r]);   // the transposed reading
          ok += memcmp(&f, &hD[l * 4 + r], 4) == 0;
          okT += memcmp(&t, &hD[l * 4 + r], 4) == 0;
          ++total;
        }
    }
    printf("v_mfma_f32_4x4x1_16B_f32 vs fmaf(a,b,c): lane=(block,col) vgpr=row: %d / %d bit-equal; transposed reading: %d / %d\n",
           ok, total, okT, total);
  }
  for (int threads : {256, 512}) {
    run<0>("16x16x4 dependent chain, registers", threads, out, cyc, seed, 2, 2048.0);
    run<4>("16x16x4 + 2 ds_read_b128 / group, no select", threads, out, cyc, seed, 2, 2048.0);
    run<1>("16x16x4 + 2 ds_read_b128 + 4 v_cndmask / group", threads, out, cyc, seed, 2, 2048.0);
    run<2>("4x4x1 dependent chain, registers", threads, out, cyc, seed, 8, 512.0);
    run<3>("4x4x1 + 4 ds_read_b128 / group", threads, out, cyc, seed, 8, 512.0);
  }
  printf("(peak fp32 matrix rate 157.3 TFLOP/s = 64 FLOP/clk/SIMD at 2.4 GHz)\n");
  return 0;
}
