// Microbenchmark: what fp32 MFMA rate does the MI355X SUSTAIN (seconds, not milliseconds), and does it depend on the
// operand data?  256 x 2 blocks of 4 waves issue nothing but v_mfma_f32_32x32x2_f32 on register operands that are
// either all zero or random normal floats; the rate is printed per window of launches.
//   hipcc --offload-arch=gfx950 -O3 -o sustain_probe sustain_probe.hip && ./sustain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void probe(const float* __restrict__ data, float* out, int iters) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float av[8], bv[8];
  for (int i = 0; i < 8; ++i) {
    av[i] = data[(i * 2 + 0) * 256 + threadIdx.x];
    bv[i] = data[(i * 2 + 1) * 256 + threadIdx.x];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[(s + a) & 7], acc[a], 0, 0, 0);
  }
  float s = 0;
  for (int a = 0; a < 4; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static void run(const char* name, const float* data, float* out, int launches, int window) {
  const int blocks = 512, iters = 60000;  // ~ 30 ms per launch
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%s:", name);
  for (int w = 0; w < launches / window; ++w) {
    hipEventRecord(e0);
    for (int i = 0; i < window; ++i) hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, 0, data, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)window * blocks * 4 * iters * 32.0 * 4096.0;
    printf(" %.1f", flops / ms / 1e9);
    fflush(stdout);
  }
  printf("  TFLOP/s per %d launches\n", window);
}

int main() {
  float *dz, *dr, *out;
  hipMalloc(&dz, 16 * 256 * 4);
  hipMalloc(&dr, 16 * 256 * 4);
  hipMalloc(&out, 512 * 256 * 4);
  hipMemset(dz, 0, 16 * 256 * 4);
  std::vector<float> h(16 * 256);
  srand(1);
  for (float& x : h) {
    const double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
    x = (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
  }
  hipMemcpy(dr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run("zero operands  ", dz, out, 60, 10);
  run("random operands", dr, out, 120, 10);
  run("zero operands  ", dz, out, 60, 10);
  return 0;
}
