// out[c] = sum_{p<P} part[p*pstride + c], c < N, routed to up to three output arrays by column
// range (c < split1 -> out0, c < split2 -> out1, else out2).  Block = COLS columns x (256/COLS)
// partial groups whose sums meet in LDS; grid = ceil(N/COLS).  Two shapes: COLS = 32 for the big
// [P][N] partial buffers (embedding tables), COLS = 8 for the few-hundred-column LayerNorm /
// bias partials, where 32-column blocks would leave the chip to 16-24 workgroups each walking
// P/8 dependent loads (measured 9 us per launch on the critical path).
#pragma once
#include "common.h"

template <int COLS>
__global__ __launch_bounds__(256) static void reduce_rows_kernel(const float* __restrict__ part,
                                                                 float* __restrict__ out0,
                                                                 float* __restrict__ out1,
                                                                 float* __restrict__ out2, long long split1,
                                                                 long long split2, int P, long long N,
                                                                 long long pstride) {
  constexpr int G = 256 / COLS;
  __shared__ float red[G][COLS + 1];
  const int col = threadIdx.x % COLS, grp = threadIdx.x / COLS;
  const long long c = (long long)blockIdx.x * COLS + col;
  // 8 independent loads in flight per thread: the kernel is a chain of dependent L2/HBM round
  // trips otherwise (P/G rows per thread, ~1 us each)
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    int p = grp;
    for (; p + 7 * G < P; p += 8 * G) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += part[(long long)(p + u * G) * pstride + c];
    }
    for (; p < P; p += G) s[0] += part[(long long)p * pstride + c];
  }
  red[grp][col] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (grp == 0 && c < N) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) s += red[g][col];
    if (c < split1) out0[c] = s;
    else if (c < split2) out1[c - split1] = s;
    else out2[c - split2] = s;
  }
}

// Several independent partial buffers in ONE launch (blockIdx.y = job): the parameter-gradient
// partials of all LayerNorm layers are summed together at the end of the backward pass instead of
// one 7 us launch per layer on the critical path.
struct ReduceJobs {
  const float* part[MFP_MAX_REDUCE_JOBS];
  float* out0[MFP_MAX_REDUCE_JOBS];
  float* out1[MFP_MAX_REDUCE_JOBS];
  float* out2[MFP_MAX_REDUCE_JOBS];
  long long split1[MFP_MAX_REDUCE_JOBS], split2[MFP_MAX_REDUCE_JOBS], N[MFP_MAX_REDUCE_JOBS], pstride[MFP_MAX_REDUCE_JOBS];
  int P[MFP_MAX_REDUCE_JOBS];
};

template <int COLS>
__global__ __launch_bounds__(256) static void reduce_rows_multi_kernel(ReduceJobs jobs) {
  constexpr int G = 256 / COLS;
  __shared__ float red[G][COLS + 1];
  const int jb = blockIdx.y;
  const long long N = jobs.N[jb], pstride = jobs.pstride[jb];
  if ((long long)blockIdx.x * COLS >= N) return;
  const float* __restrict__ part = jobs.part[jb];
  const int P = jobs.P[jb];
  const int col = threadIdx.x % COLS, grp = threadIdx.x / COLS;
  const long long c = (long long)blockIdx.x * COLS + col;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    int p = grp;
    for (; p + 7 * G < P; p += 8 * G) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s[u] += part[(long long)(p + u * G) * pstride + c];
    }
    for (; p < P; p += G) s[0] += part[(long long)p * pstride + c];
  }
  red[grp][col] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (grp == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) t += red[g][col];
    if (c < jobs.split1[jb]) jobs.out0[jb][c] = t;
    else if (c < jobs.split2[jb]) jobs.out1[jb][c - jobs.split1[jb]] = t;
    else jobs.out2[jb][c - jobs.split2[jb]] = t;
  }
}

static inline void launch_reduce_rows3(const float* part, float* out0, float* out1, float* out2, long long split1,
                                       long long split2, int P, long long N, long long pstride, hipStream_t st) {
  if (N <= 8192)
    hipLaunchKernelGGL(reduce_rows_kernel<16>, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, st, part, out0, out1, out2,
                       split1, split2, P, N, pstride);
  else
    hipLaunchKernelGGL(reduce_rows_kernel<32>, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, st, part, out0, out1,
                       out2, split1, split2, P, N, pstride);
}
static inline void launch_reduce_rows(const float* part, float* out0, float* out1, long long split, int P,
                                      long long N, long long pstride, hipStream_t st) {
  launch_reduce_rows3(part, out0, out1, out1, split, N, P, N, pstride, st);
}
