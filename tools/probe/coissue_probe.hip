// Microbenchmark: can the fp32 vector pipe (v_pk_fma_f32) add FLOPs beside a saturated fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32) on the same SIMD?  Each wave runs 4 independent MFMA accumulators back to back with NV
// packed FMAs (independent accumulators, register operands) behind every MFMA.
//   hipcc --offload-arch=gfx950 -O3 -o coissue_probe coissue_probe.hip && ./coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NV>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  extern __shared__ unsigned char smem[];
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  f32x2 vacc[16];
  for (int i = 0; i < 16; ++i) vacc[i] = f32x2{0.f, 0.f};
  float av = threadIdx.x * 0.001f, bv = 1.0f;
  f32x2 va = {av, av + 1.f}, vb = {0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[a]) : "v"(av), "v"(bv));
#pragma unroll
        for (int v = 0; v < NV; ++v)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(vacc[v]) : "v"(va), "v"(vb));
      }
    }
  }
  float s = 0;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int i = 0; i < 16; ++i) s += vacc[i][0] + vacc[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(int blocks, int lds_bytes, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipFuncSetAttribute((const void*)probe<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<NV>, dim3(blocks), dim3(256), lds_bytes, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<NV>, dim3(blocks), dim3(256), lds_bytes, 0, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * 4;
  const double mf = waves * iters * 16.0 * 4096.0, vf = waves * iters * 16.0 * NV * 64.0 * 4.0;
  printf("NV %2d  blocks/CU %d  %.3f ms  MFMA %.1f TF  + VALU %.1f TF  = %.1f TF\n", NV, blocks / 256, ms, mf / ms / 1e9,
         vf / ms / 1e9, (mf + vf) / ms / 1e9);
  hipFree(out);
}

int main() {
  const int it = 10000;
  for (int bpc = 1; bpc <= 2; ++bpc) {
    const int blocks = 256 * bpc, lds = bpc == 1 ? 150 * 1024 : 68 * 1024;
    run<0>(blocks, lds, it);
    run<1>(blocks, lds, it);
    run<2>(blocks, lds, it);
    run<4>(blocks, lds, it);
    run<6>(blocks, lds, it);
    run<8>(blocks, lds, it);
    run<12>(blocks, lds, it);
    run<16>(blocks, lds, it);
  }
  return 0;
}
