// out[c] = sum_{p<P} part[p*pstride + c], c < N.  Block = 32 columns x 8 partial groups; the
// 8 group sums meet in LDS.  Grid = ceil(N/32): full-chip parallel for the [P][N] partial
// buffers of the LN / colsum / embedding backward kernels (all a few MB at most).
#pragma once
#include "common.h"

__global__ __launch_bounds__(256) static void reduce_rows_kernel(const float* __restrict__ part,
                                                                 float* __restrict__ out0,
                                                                 float* __restrict__ out1, int split,
                                                                 int P, long long N, long long pstride) {
  __shared__ float red[8][33];
  const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long long c = (long long)blockIdx.x * 32 + col;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < N) {
    int p = grp;
    for (; p + 24 < P; p += 32) {
      s0 += part[(long long)p * pstride + c];
      s1 += part[(long long)(p + 8) * pstride + c];
      s2 += part[(long long)(p + 16) * pstride + c];
      s3 += part[(long long)(p + 24) * pstride + c];
    }
    for (; p < P; p += 8) s0 += part[(long long)p * pstride + c];
  }
  red[grp][col] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (grp == 0 && c < N) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += red[g][col];
    if (c < split) out0[c] = s; else out1[c - split] = s;
  }
}

static inline void launch_reduce_rows(const float* part, float* out0, float* out1, long long split, int P,
                                      long long N, long long pstride, hipStream_t st) {
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, st, part, out0, out1,
                     (int)split, P, N, pstride);
}
