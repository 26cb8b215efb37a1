// Is v_mfma_f32_16x16x4_f32 an exact fp32 fma chain over its four k values, in k order, like the 32x32x2 instruction
// the conv kernel's bit-exactness rests on?  (A 16x16 wave tile would quarter the fma chain of a lone wave at batch 1.)
// Random normal A (16x4), B (4x16), C (16x16); the device result is compared bit for bit with
// fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, c)))) and with the reversed chain.
//   hipcc --offload-arch=gfx950 -O3 -o mfma16_order_probe mfma16_order_probe.hip && ./mfma16_order_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, const float* C, float* D) {
  const int l = threadIdx.x;                 // one wave
  const float a = A[(l % 16) * 4 + l / 16];  // A[i][k]: lane holds i = l % 16, k = l / 16
  const float b = B[(l / 16) * 16 + l % 16]; // B[k][j]: lane holds k = l / 16, j = l % 16
  f32x4 c;
  for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l / 16) + r) * 16 + l % 16];  // D[i = 4*(l/16) + r][j = l % 16]
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = c[r];
}

static float rnd() {
  double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
  return (float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
}

int main() {
  float hA[64], hB[64], hC[256], hD[256], *dA, *dB, *dC, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
  int fwd = 0, rev = 0, total = 0;
  srand(7);
  for (int trial = 0; trial < 200; ++trial) {
    for (float& x : hA) x = rnd();
    for (float& x : hB) x = rnd();
    for (float& x : hC) x = rnd() * (trial % 3 ? 1.0f : 1e-3f);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        float f = hC[i * 16 + j], r = hC[i * 16 + j];
        for (int k = 0; k < 4; ++k) f = fmaf(hA[i * 4 + k], hB[k * 16 + j], f);
        for (int k = 3; k >= 0; --k) r = fmaf(hA[i * 4 + k], hB[k * 16 + j], r);
        fwd += memcmp(&f, &hD[i * 16 + j], 4) == 0;
        rev += memcmp(&r, &hD[i * 16 + j], 4) == 0;
        ++total;
      }
  }
  printf("v_mfma_f32_16x16x4_f32 vs fmaf chain: k ascending %d / %d bit-equal, k descending %d / %d\n", fwd, total, rev, total);
  return 0;
}
