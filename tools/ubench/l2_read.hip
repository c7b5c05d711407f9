// Microbenchmark: aggregate read bandwidth of L2/MALL-resident data with the GEMM's access pattern
// (each wave instruction = 8 rows x 128 B, 16 B per lane), as a function of footprint.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__global__ __launch_bounds__(256) void k(const unsigned char* __restrict__ base, size_t footprint, int iters,
                                         int row_stride, unsigned int* out) {
  const int tid = threadIdx.x;
  // a "tile load" = 256 threads x 16 B: 32 rows x 128 B, rows row_stride bytes apart
  const size_t tile_bytes = (size_t)32 * row_stride;
  u32x4 acc = {0, 0, 0, 0};
  size_t off = ((size_t)blockIdx.x * 7919u * tile_bytes) % (footprint - 8 * tile_bytes);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // 8 independent 16-B loads per thread in flight (32 KB per WG)
      const u32x4 v = *reinterpret_cast<const u32x4*>(base + off + (size_t)j * tile_bytes + (size_t)(tid >> 3) * row_stride + (tid & 7) * 16);
      acc += v;
    }
    off += 8 * tile_bytes;
    if (off >= footprint - 8 * tile_bytes) off = 0;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345u) out[0] = 1;
}
int main() {
  unsigned char* buf; unsigned int* out;
  const size_t maxfp = (size_t)1 << 30;
  hipMalloc(&buf, maxfp); hipMemset(buf, 1, maxfp); hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 64;
  for (size_t fp : {(size_t)1 << 20, (size_t)8 << 20, (size_t)24 << 20, (size_t)128 << 20, (size_t)1 << 30}) {
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, buf, fp, iters, 512, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("footprint %5zu MB  WGs %4d: %.2f TB/s (%.1f us)\n", fp >> 20, wgs,
                        (double)wgs * iters * 8 * 4096 / (ms * 1e-3) / 1e12, ms * 1e3);
      }
    }
  }
  return 0;
}
