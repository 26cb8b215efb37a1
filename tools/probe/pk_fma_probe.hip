// Rate of v_pk_fma_f32 with a SCALAR pair as the multiplier (op_sel broadcasts one of its halves to both results), the
// form a vector-pipe convolution would use: acc pair = two output channels of one pixel, src1 = their two weights,
// src0 = the pixel's activation (scalar).  Registers only; R independent accumulator pairs per lane.
//   hipcc --offload-arch=gfx950 -O3 -o pk_fma_probe pk_fma_probe.hip && ./pk_fma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 8192;

template <int R, int MODE>  // MODE 0: v_pk_fma s-pair, v, v   1: v_pk_fma v, v, v   2: v_fmac s, v (plain)
__global__ void bench(const float* seed, float* out) {
  const int lane = threadIdx.x & 63;
  f32x2 acc[R], w[4];
  float sacc[2 * R];
  for (int i = 0; i < R; ++i) acc[i] = f32x2{0.f, 0.f};
  for (int i = 0; i < 2 * R; ++i) sacc[i] = 0.f;
  for (int i = 0; i < 4; ++i) w[i] = f32x2{seed[lane * 8 + 2 * i], seed[lane * 8 + 2 * i + 1]};
  f32x2 sa[4];
  for (int i = 0; i < 4; ++i) asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(sa[i]) : "s"(seed), "s"(i * 8));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sa[0]), "+s"(sa[1]), "+s"(sa[2]), "+s"(sa[3]));
  f32x2 va[4];
  for (int i = 0; i < 4; ++i) va[i] = f32x2{seed[64 + i], seed[80 + i]};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        if (MODE == 0)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"(sa[(k + i) & 3]), "v"(w[k]));
        else if (MODE == 1)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va[(k + i) & 3]), "v"(w[k]));
        else {
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sacc[2 * i]) : "s"(sa[(k + i) & 3][0]), "v"(w[k][0]));
          asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(sacc[2 * i + 1]) : "s"(sa[(k + i) & 3][1]), "v"(w[k][1]));
        }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < R; ++i) s += acc[i][0] + acc[i][1] + sacc[2 * i] + sacc[2 * i + 1];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, int MODE>
static void run(const char* name, int threads, int bpc, const float* seed, float* out) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((bench<R, MODE>), dim3(256 * bpc), dim3(threads), 0, 0, seed, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((bench<R, MODE>), dim3(256 * bpc), dim3(threads), 0, 0, seed, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * bpc * (threads / 64) * ITERS * 4.0 * R * 64 * 4;  // 2 fma = 4 flop per lane per step
  printf("%-30s R=%d  %d waves/SIMD: %7.3f ms  %6.1f TFLOP/s\n", name, R, threads / 256 * bpc, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *seed, *out;
  (void)hipMalloc(&seed, 4096); (void)hipMalloc(&out, 256 * 4 * 1024 * 4);
  float h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 1e-3f * (i % 7);
  (void)hipMemcpy(seed, h, 4096, hipMemcpyHostToDevice);
  for (int threads : {256, 512, 1024}) {
    run<4, 0>("v_pk_fma_f32 s-pair (op_sel)", threads, 1, seed, out);
    run<8, 0>("v_pk_fma_f32 s-pair (op_sel)", threads, 1, seed, out);
    run<4, 1>("v_pk_fma_f32 v, v, v", threads, 1, seed, out);
    run<8, 1>("v_pk_fma_f32 v, v, v", threads, 1, seed, out);
    run<4, 2>("v_fmac_f32 s, v", threads, 1, seed, out);
  }
  return 0;
}
