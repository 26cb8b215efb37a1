// Microbenchmark: cost of the LDS-DMA issue inside the conv kernel's k-loop, and whether a dedicated
// loader wave removes it.  Block = 4 compute waves (+1 loader wave in MODE 2); per "k-step" each
// compute wave does 4 x [4 ds_read_b128 -> 16 MFMA]; 32 LDS-DMA instructions (1 KiB each) refill the other
// half of a 64 KiB double buffer; one barrier per k-step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void lds_dma16(const float* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory", "m0");
}

// MODE 0: no DMA.  1: every compute wave issues 8 DMA per k-step (2 per MFMA group).  2: a fifth wave issues all 32.
template <int MODE>
__global__ __launch_bounds__(MODE == 2 ? 320 : 256) void probe(const float* src, float* out, int ksteps, long long stream_mask) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, lrow = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const float* gp = src + ((size_t)blockIdx.x * 64 + lane) * 4;  // small, cache-resident source
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  if (MODE == 2 && wave == 4) {
    for (int t = 0; t < ksteps; ++t) {
      const unsigned base = lds0 + ((t + 1) & 1) * 32768;
#pragma unroll
      for (int j = 0; j < 32; ++j) lds_dma16(gp + j * 256, base + j * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    return;
  }
  for (int t = 0; t < ksteps; ++t) {
    const unsigned char* buf = smem + (t & 1) * 32768;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = j * 32 + lrow;
        q[j] = *reinterpret_cast<const f32x4*>(buf + row * 128 + (((2 * g + h) ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[0][s], q[2][s], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[0][s], q[3][s], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[1][s], q[2][s], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(q[1][s], q[3][s], acc[3], 0, 0, 0);
        if (MODE == 1 && s == 0) {
          const unsigned base = lds0 + ((t + 1) & 1) * 32768 + (g * 8 + wave * 2) * 1024;
          // stream_mask != 0: every k-step reads a fresh 32 KiB of a large buffer (real L2/HBM traffic)
          const long long so = (((long long)blockIdx.x * 977 + t) * 8192LL + (g * 8 + wave * 2) * 256) & stream_mask;
          lds_dma16(gp + so, base);
          lds_dma16(gp + so + 256, base + 1024);
        }
      }
    }
    if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + tid] = s;
}

// MODE 3: the same work as MODE 1 but the 64 KiB are a ring of four 16-k half-tiles (16 KiB each: 128 rows x 64 B
// for A and for W); each half-tile step = 2 groups of [4 ds_read_b128 -> 16 MFMA] and 4 LDS-DMA instructions per
// wave that refill the stage consumed one step earlier (prefetch distance 3 stages), counted vmcnt(8).
__global__ __launch_bounds__(256) void probe_ring(const float* src, float* out, int ksteps, long long stream_mask) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[65536];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, lrow = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const unsigned lds0 = (unsigned)(unsigned long long)LDS_PTR(smem);
  const float* gp = src + ((size_t)blockIdx.x * 64 + lane) * 4;
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const int nst = ksteps * 2;  // half-tile steps
  auto dma = [&](int st) {     // 4 instructions per wave refill stage (st & 3)
    const unsigned base = lds0 + (st & 3) * 16384 + wave * 4096;
    const long long so = (((long long)blockIdx.x * 977 + st) * 4096LL + wave * 1024) & stream_mask;
#pragma unroll
    for (int j = 0; j < 4; ++j) lds_dma16(gp + so + j * 256, base + j * 1024);
  };
  dma(0); dma(1); dma(2);
  for (int st = 0; st < nst; ++st) {
    // stage st must have landed: everything but the 2 younger stages (8 instructions) is complete
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    if (st + 3 < nst) dma(st + 3); else { dma(st & 3); }  // keep the instruction count constant for the counted wait
    const unsigned char* buf = smem + (st & 3) * 16384;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      f32x4 q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = j * 32 + lrow;
        q[j] = *reinterpret_cast<const f32x4*>(buf + (j >> 1) * 8192 + (row & 63) * 64 + ((((2 * g + h) ^ ((row >> 2) & 3))) << 4));
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
  out[blockIdx.x * 256 + tid] = s;
}

void run_ring(const char* name, int blocks, int ksteps, long long stream_floats) {
  float *out, *src;
  hipMalloc(&out, (size_t)blocks * 320 * 4);
  const size_t src_bytes = (size_t)blocks * 64 * 16 + 65536 * 4 + (size_t)stream_floats * 4;
  hipMalloc(&src, src_bytes);
  hipMemset(src, 0, src_bytes);
  const long long mask = stream_floats ? stream_floats - 1 : 0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe_ring, dim3(blocks), dim3(256), 0, 0, src, out, ksteps, mask);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe_ring, dim3(blocks), dim3(256), 0, 0, src, out, ksteps, mask);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * ksteps * 64.0 * 4096.0;
  printf("%-52s blocks %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
  hipFree(out); hipFree(src);
}

template <int MODE>
void run(const char* name, int blocks, int ksteps, long long stream_floats = 0) {
  float *out, *src;
  hipMalloc(&out, (size_t)blocks * 320 * 4);
  const size_t src_bytes = (size_t)blocks * 64 * 16 + 65536 * 4 + (size_t)stream_floats * 4;
  hipMalloc(&src, src_bytes);
  hipMemset(src, 0, src_bytes);
  const long long mask = stream_floats ? stream_floats - 1 : 0;
  const int threads = MODE == 2 ? 320 : 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, src, out, ksteps, mask);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, src, out, ksteps, mask);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * ksteps * 64.0 * 4096.0;
  printf("%-52s blocks %5d  %.3f ms  %.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
  hipFree(out); hipFree(src);
}

int main() {
  const int ks = 4000;
  run<0>("no DMA (2 blocks/CU)", 512, ks);
  run<1>("DMA by compute waves, interleaved (2 blocks/CU)", 512, ks);
  run<2>("DMA by a 5th loader wave (2 blocks/CU)", 512, ks);
  run<0>("no DMA, 3 rounds", 1536, ks / 2);
  run<1>("DMA by compute waves, 3 rounds", 1536, ks / 2);
  run<2>("DMA by loader wave, 3 rounds", 1536, ks / 2);
  run<1>("DMA by compute waves, streaming 64 MiB (MALL)", 512, ks, 1LL << 24);
  run<1>("DMA by compute waves, streaming 2 GiB (HBM)", 512, ks, 1LL << 29);
  run_ring("4-stage ring, cached source", 512, ks, 0);
  run_ring("4-stage ring, streaming 64 MiB (MALL)", 512, ks, 1LL << 24);
  run_ring("4-stage ring, streaming 2 GiB (HBM)", 512, ks, 1LL << 29);
  return 0;
}
