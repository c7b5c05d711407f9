// Microbenchmark: LDS atomic throughput on gfx950 (f32 add vs u32 add vs u64 add vs plain RMW).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ rows, float* out, int iters) {
  __shared__ unsigned long long tab[128 * 64];  // 64 KB
  float* tf = (float*)tab; unsigned int* tu = (unsigned int*)tab;
  for (int i = threadIdx.x; i < 128 * 64; i += 256) tab[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
    int row = rows[(blockIdx.x * 4 + wave) * iters + it];  // uniform per wave
    if (MODE == 0) atomicAdd(&tf[row * 64 + lane], g);
    if (MODE == 1) atomicAdd(&tu[row * 64 + lane], (unsigned int)lane);
    if (MODE == 2) atomicAdd(&tab[row * 64 + lane], (unsigned long long)lane);
    if (MODE == 3) { float v = tf[(wave * 32 + (row & 31)) * 64 + lane]; tf[(wave * 32 + (row & 31)) * 64 + lane] = v + g; }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tf[5] + (float)tu[7];
}
int main() {
  const int blocks = 512, iters = 2048;
  int* rows; float* out;
  hipMalloc(&rows, blocks * 4 * iters * sizeof(int)); hipMalloc(&out, blocks * sizeof(float));
  int* h = (int*)malloc(blocks * 4 * iters * sizeof(int));
  for (int i = 0; i < blocks * 4 * iters; ++i) h[i] = rand() % 128;
  hipMemcpy(rows, h, blocks * 4 * iters * sizeof(int), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain RMW f32 (wave-private)"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, rows, out, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, rows, out, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, rows, out, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, rows, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) printf("%-30s %8.1f us  -> %.2f G wave-ops/s, %.1f cycles/wave-op/CU@2.4GHz (2 WG/CU)\n", names[mode], ms * 1e3,
                           (double)blocks * 4 * iters / (ms * 1e-3) / 1e9, ms * 1e-3 * 2.4e9 / ((double)blocks * 4 * iters / 256));
    }
  }
  return 0;
}
