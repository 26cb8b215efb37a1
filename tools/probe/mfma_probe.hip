// Microbenchmark: what does the fp32 MFMA pipe sustain under the conv kernel's instruction mix?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, int lds_pad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, h = lane >> 5, lrow = lane & 31;
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av = lane * 0.001f, bv = 1.0f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // MFMA only, 16 per iteration
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
    } else {                    // 4 ds_read_b128 -> wait -> 16 MFMA (the conv kernel's group)
      f32x4 q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = j * 32 + lrow;
        q[j] = *reinterpret_cast<const f32x4*>(smem + row * 128 + ((((it & 3) * 2 + h) ^ ((row >> 1) & 7)) << 4));
      }
      if (MODE == 2) __syncthreads();
      if (MODE == 3) {  // the conv kernel's permlane32_swap exchange
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const u32x2 s01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[j][0]), __float_as_uint(q[j][1]), false, false);
          const u32x2 s23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(q[j][2]), __float_as_uint(q[j][3]), false, false);
          q[j][0] = __uint_as_float(s01[0]);
          q[j][2] = __uint_as_float(s01[1]);
          q[j][1] = __uint_as_float(s23[0]);
          q[j][3] = __uint_as_float(s23[1]);
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[0][s], q[2][s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[0][s], q[3][s], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[1][s], q[2][s], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[1][s], q[3][s], acc[3], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, int lds_bytes, int iters) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), lds_bytes, 0, out, iters, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), lds_bytes, 0, out, iters, 0);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 /*waves*/ * iters * 16.0 * 4096.0;
  printf("%-44s blocks %5d lds %6d  %.3f ms  %.1f TFLOP/s\n", name, blocks, lds_bytes, ms, flops / ms / 1e9);
  hipFree(out);
}

int main() {
  const int it = 20000;
  // blocks per CU controlled by LDS size: 160K -> 1/CU, 68K -> 2/CU, 40K -> 4/CU (cap by waves 8)
  run<0>("mfma only, 1 block/CU (1 wave/SIMD)", 256, 150 * 1024, it);
  run<0>("mfma only, 2 blocks/CU", 512, 68 * 1024, it);
  run<0>("mfma only, 4 blocks/CU", 1024, 36 * 1024, it);
  run<1>("lds+mfma, 1 block/CU", 256, 150 * 1024, it);
  run<1>("lds+mfma, 2 blocks/CU", 512, 68 * 1024, it);
  run<1>("lds+mfma, 3 blocks/CU", 768, 50 * 1024, it);
  run<1>("lds+mfma, 4 blocks/CU", 1024, 36 * 1024, it);
  run<2>("lds+barrier+mfma, 2 blocks/CU", 512, 68 * 1024, it);
  run<2>("lds+barrier+mfma, 4 blocks/CU", 1024, 36 * 1024, it);
  run<3>("lds+permlane+mfma, 2 blocks/CU", 512, 68 * 1024, it);
  run<3>("lds+permlane+mfma, 1 block/CU", 256, 150 * 1024, it);
  return 0;
}
