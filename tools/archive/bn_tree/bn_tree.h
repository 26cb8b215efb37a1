// bn_tree.h — device side of BnTreeArgs (kernels.h): the last workgroup to arrive finishes a statistics reduction.
//
// Level-0 rows [row][2][C] (T = double: the per-tile sums of the train-forward conv; float: the per-chunk sums of
// bn_bwd_reduce) are written by the workgroups of ONE launch, each row x column block by exactly one workgroup.  After its
// row a workgroup calls vy_bn_tree_finish: it counts itself into its group of g1 consecutive rows; the last arrival of a
// group adds the group's rows in row order (double) into the level-1 row, counts the group in, and the last group's arrival
// adds the level-1 rows in group order and hands the two totals of every channel to `fin`.  The summation tree is fixed by
// (rows, g1) alone — which workgroup happens to be last changes nothing — so a training step stays bit-reproducible.
//
// Visibility between workgroups (possibly on different XCDs, each with its own L2): rows are stored with agent-scope atomic
// stores (write-through, `sc1`), followed by s_waitcnt vmcnt(0) and the workgroup barrier BEFORE the counter's atomic add;
// the finisher reads with agent-scope atomic loads (past its L1 / L2) AFTER its own add returned the last ticket — the same
// hand-off as the split-K reduce of conv_igemm.hip.  Counters are reset by the finisher: zero again when the launch ends.
#pragma once
#include "kernels.h"

template <typename T>
__device__ __forceinline__ void vy_store_agent(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// p[0], p[stride], ... p[(n - 1) * stride] added in that order to +0.  The loads go past the caches: one round trip to
// memory each, so up to 32 are in flight at a time (rows beyond n read row n - 1 again and are not added: no branches
// between the loads) — a tree level of <= 32 rows costs ONE round trip.
template <typename T>
__device__ __forceinline__ double vy_ordered_sum_agent(const T* p, long long stride, int n) {
  double s = 0.0;
  for (int r = 0; r < n; r += 32) {
    T v[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int rr = r + u < n ? r + u : n - 1;
      v[u] = __hip_atomic_load(p + (long long)rr * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int u = 0; u < 32; ++u)
      if (r + u < n) s += (double)v[u];
  }
  return s;
}

// batch statistics -> normalisation coefficients (mxnet BatchNorm, train mode; layers.py:68)
__device__ __forceinline__ void bn_finalize_channel(const BnFinalizeArgs& a, int c, double sum1, double sum2) {
  const double mean = sum1 / a.count;
  double var = sum2 / a.count - mean * mean;  // biased (mxnet BatchNorm)
  if (var < 0.0) var = 0.0;
  const float mf = (float)mean, vf = (float)var;
  const float invstd = 1.0f / sqrtf(vf + a.eps);
  const float sc = a.gamma[c] * invstd;
  a.scale[c] = sc;
  a.shift[c] = fmaf(-mf, sc, a.beta[c]);
  a.save_mean[c] = mf;
  a.save_invstd[c] = invstd;
  a.running_mean[c] = a.running_mean[c] * a.momentum + mf * (1.0f - a.momentum);
  a.running_var[c] = a.running_var[c] * a.momentum + vf * (1.0f - a.momentum);
}

// Called by EVERY thread of the workgroup (tid = 0 .. nt - 1, nt >= 2 * ncols) after the workgroup's level-0 row `row`
// (columns col0 .. col0 + ncols - 1 of C, column block `colblock`) was stored with vy_store_agent.  lds: 16 bytes +
// 2 * ncols doubles of LDS nobody else is using any more.  fin(c, s1, s2): called for the channels of this column block by
// the ONE workgroup that completes the launch's last group.
template <typename T, typename Fin>
__device__ __forceinline__ void vy_bn_tree_finish(const BnTreeArgs& bt, const T* rows0, int C, int col0, int ncols, int colblock,
                                                  int row, int n_rows, int tid, unsigned char* lds, Fin fin) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's row is written through
  __syncthreads();
  volatile int* lflag = reinterpret_cast<volatile int*>(lds);
  double* red2 = reinterpret_cast<double*>(lds + 16);
  const int g = row / bt.g1;
  const int first = g * bt.g1;
  const int nr = bt.g1 < n_rows - first ? bt.g1 : n_rows - first;
  unsigned* cnt = bt.cnt + (long long)colblock * (bt.groups + 1);
  if (tid == 0)
    lflag[0] = __hip_atomic_fetch_add(cnt + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nr - 1) ? 1 : 0;
  __syncthreads();
  if (!lflag[0]) return;
  const int kind = tid >= ncols ? 1 : 0, col = tid - kind * ncols;
  const bool mine = tid < 2 * ncols && col0 + col < C;
  if (mine) {
    const double s = vy_ordered_sum_agent(rows0 + ((long long)first * 2 + kind) * C + col0 + col, 2LL * C, nr);
    vy_store_agent(bt.rows + ((long long)g * 2 + kind) * C + col0 + col, s);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // (also: every thread has read lflag[0])
  if (tid == 0) {
    __hip_atomic_store(cnt + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lflag[0] = __hip_atomic_fetch_add(cnt + bt.groups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(bt.groups - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!lflag[0]) return;
  if (mine) red2[tid] = vy_ordered_sum_agent(bt.rows + (long long)kind * C + col0 + col, 2LL * C, bt.groups);
  __syncthreads();
  if (tid < ncols && col0 + tid < C) fin(col0 + tid, red2[tid], red2[ncols + tid]);
  if (tid == 0) __hip_atomic_store(cnt + bt.groups, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
