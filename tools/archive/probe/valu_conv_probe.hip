// Can the VECTOR pipe carry a small-batch convolution at the matrix rate?  On gfx950 fp32 peaks are equal (157.3 TFLOP/s
// vector = 157.3 fp32 MFMA: the fp32 MFMA runs on the vector FMA hardware), and a v_fma_f32 chain per lane is the
// finest possible work unit: lane = output channel, a few registers = output pixels, the activation value comes in as a
// SCALAR operand (wave-uniform: s_load from the pixel's channel vector), the weight from LDS.  This probe times exactly
// that inner loop — per 8-channel group and wave: R s_load_dwordx8 (activations of R pixels), 2 ds_read_b128 (the
// lane's weights), 8 R v_fma_f32 in the pinned order 0,4,1,5,2,6,3,7 — with 1 / 2 / 4 waves per SIMD on every CU,
// software-pipelined one group ahead or not.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_conv_probe valu_conv_probe.hip && ./valu_conv_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

constexpr int GROUPS = 4096;  // 8-channel groups per wave

template <int R, bool PIPE, int WHAT>  // WHAT bit 0: scalar loads, bit 1: LDS reads (0 = registers only)
__global__ void bench(const float* __restrict__ act, float* out, const float* seed) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 32];  // one weight tile: 64 channels x 32 k
  for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) lds[i] = seed[i & 1023];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float acc[R];
#pragma unroll
  for (int i = 0; i < R; ++i) acc[i] = 0.f;
  // wave-uniform pixel pointers (R pixels of 2048 channels each, re-read round and round: scalar-cache resident)
  const float* pa[R];
#pragma unroll
  for (int i = 0; i < R; ++i) pa[i] = act + ((blockIdx.x * 16 + wave) * R + i) % 512 * 2048;
  const float* wrow = lds + lane * 32;
  f32x8 sa[2][R];
  f32x4 wl[2], wh[2];
  for (int b = 0; b < 2; ++b) {  // defined values for the variants that skip a load kind in the loop
#pragma unroll
    for (int i = 0; i < R; ++i)
      asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(sa[b][i]) : "s"(pa[i]), "s"(b * 32));
    asm volatile("ds_read_b128 %0, %1" : "=v"(wl[b]) : "v"((unsigned)(unsigned long long)(wrow + b * 8)));
    asm volatile("ds_read_b128 %0, %1" : "=v"(wh[b]) : "v"((unsigned)(unsigned long long)(wrow + b * 8 + 4)));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  auto issue = [&](int buf, int g) {
    const int cg = g & 255;  // channel group inside the pixel vector
    if (WHAT & 1) {
#pragma unroll
      for (int i = 0; i < R; ++i)
        asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(sa[buf][i]) : "s"(pa[i]), "s"(cg * 32));
    }
    if (WHAT & 2) {
      const int c4 = (g & 3) * 2;
      const int sw = (lane >> 1) & 7;
      asm volatile("ds_read_b128 %0, %1" : "=v"(wl[buf]) : "v"((unsigned)(unsigned long long)(wrow + (((c4) ^ sw) << 2))));
      asm volatile("ds_read_b128 %0, %1" : "=v"(wh[buf]) : "v"((unsigned)(unsigned long long)(wrow + (((c4 + 1) ^ sw) << 2))));
    }
  };
  auto wait = [&](int buf) {
    if (R == 4)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sa[buf][0]), "+s"(sa[buf][1]), "+s"(sa[buf][R > 2 ? 2 : 0]), "+s"(sa[buf][R > 3 ? 3 : 0]),
                   "+v"(wl[buf]), "+v"(wh[buf]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(sa[buf][0]), "+s"(sa[buf][R > 1 ? 1 : 0]), "+v"(wl[buf]), "+v"(wh[buf]));
  };
  auto fmas = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int i = 0; i < R; ++i) acc[i] = __builtin_fmaf(sa[buf][i][kk], wl[buf][kk], acc[i]);
#pragma unroll
      for (int i = 0; i < R; ++i) acc[i] = __builtin_fmaf(sa[buf][i][4 + kk], wh[buf][kk], acc[i]);
    }
  };
  if (PIPE) {
    issue(0, 0);
    for (int g = 0; g < GROUPS; g += 2) {
      wait(0);
      issue(1, g + 1);
      fmas(0);
      wait(1);
      issue(0, g + 2);
      fmas(1);
    }
    wait(0);
  } else {
    for (int g = 0; g < GROUPS; ++g) {
      issue(0, g);
      wait(0);
      fmas(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < R; ++i) s += acc[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R, bool PIPE, int WHAT>
static void run(int threads, int blocks_per_cu, const float* act, float* out, const float* seed) {
  const int blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((bench<R, PIPE, WHAT>), dim3(blocks), dim3(threads), 0, 0, act, out, seed);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((bench<R, PIPE, WHAT>), dim3(blocks), dim3(threads), 0, 0, act, out, seed);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * (threads / 64) * GROUPS * 8.0 * R * 64 * 2;
  printf("R=%d %s %s%s  %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s\n", R, PIPE ? "pipelined " : "load-wait-use", WHAT & 1 ? "+smem" : "     ", WHAT & 2 ? "+lds" : "    ", threads / 256 * blocks_per_cu, ms,
         flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *act, *out, *seed;
  (void)hipMalloc(&act, 512 * 2048 * 4);
  (void)hipMalloc(&out, 256 * 4 * 512 * 4);
  (void)hipMalloc(&seed, 4096);
  std::vector<float> h(512 * 2048);
  srand(1);
  for (float& x : h) x = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
  (void)hipMemcpy(act, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(seed, h.data(), 4096, hipMemcpyHostToDevice);
  for (int threads : {256, 512}) {
    for (int bpc : {1, 2}) {
      run<4, true, 0>(threads, bpc, act, out, seed);
      run<4, true, 1>(threads, bpc, act, out, seed);
      run<4, true, 2>(threads, bpc, act, out, seed);
      run<4, true, 3>(threads, bpc, act, out, seed);
    }
  }
  printf("(peak fp32 vector rate 157.3 TFLOP/s = 64 FLOP/clk/SIMD at 2.4 GHz)\n");
  return 0;
}
