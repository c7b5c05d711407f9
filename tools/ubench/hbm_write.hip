// Microbenchmark: achievable WRITE bandwidth on MI355X for the store patterns of the GEMM epilogue.
//  mode 0: lane-linear 16 B stores (a wave instruction = 1 KB contiguous)                 -- ideal
//  mode 1: GEMM epilogue, bf16: instruction = 16 rows x 4 pieces of 16 B at a 32 B stride;
//          a second instruction fills the other 16 B of each 32 B (rows `ld` bytes apart)
//  mode 2: GEMM epilogue, f32: instruction = 16 rows x 4 pieces of 16 B at a 64 B stride; 4 instr.
//  mode 3: like 1 but each lane issues its two 16 B pieces back to back as ONE 32 B row segment
//          via two stores to adjacent addresses (same as 1, different instruction order) -- control
//  mode 4: full-row stores: instruction = 8 rows x 128 B contiguous (what an LDS-staged epilogue does)
// plus a copy (read+write) reference.  Prints GB/s of bytes written.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ __launch_bounds__(256) void wk(unsigned char* __restrict__ base, size_t total, int ld) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lg = lane >> 4;
  const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
  if (MODE == 0) {
    for (size_t off = ((size_t)blockIdx.x * 256 + tid) * 16; off < total; off += (size_t)gridDim.x * 256 * 16)
      *reinterpret_cast<u32x4*>(base + off) = v;
  } else if (MODE == 1 || MODE == 3) {
    // tile = 32 rows x (4 waves x 128 B); matrix row = ld bytes; tiles walk down the rows, then right
    const int tiles_n = ld / 512;
    const size_t rows = total / ld, tiles_m = rows / 32;
    for (size_t t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
      const size_t tm = t / tiles_n, tn = t % tiles_n;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        unsigned char* p = base + (tm * 32 + a * 16 + li) * (size_t)ld + tn * 512 + wave * 128 + lg * 32;
        *reinterpret_cast<u32x4*>(p) = v;
        *reinterpret_cast<u32x4*>(p + 16) = v;
      }
    }
  } else if (MODE == 2) {
    const int tiles_n = ld / 1024;
    const size_t rows = total / ld, tiles_m = rows / 32;
    for (size_t t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
      const size_t tm = t / tiles_n, tn = t % tiles_n;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        unsigned char* p = base + (tm * 32 + a * 16 + li) * (size_t)ld + tn * 1024 + wave * 256 + lg * 64;
#pragma unroll
        for (int b = 0; b < 4; ++b) *reinterpret_cast<u32x4*>(p + 16 * b) = v;
      }
    }
  } else if (MODE == 4) {
    const int tiles_n = ld / 512;
    const size_t rows = total / ld, tiles_m = rows / 32;
    for (size_t t = blockIdx.x; t < tiles_m * tiles_n; t += gridDim.x) {
      const size_t tm = t / tiles_n, tn = t % tiles_n;
#pragma unroll
      for (int a = 0; a < 2; ++a) {   // wave w: rows 8w..8w+7 of each 16... full 512 B row segments: 32 lanes per row
        unsigned char* p = base + (tm * 32 + a * 16 + wave * 4 + (lane >> 5) * 2) * (size_t)ld + tn * 512 + (lane & 31) * 16;
        *reinterpret_cast<u32x4*>(p) = v;
        *reinterpret_cast<u32x4*>(p + ld) = v;
      }
    }
  }
}
__global__ __launch_bounds__(256) void copyk(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
template <int MODE> void run(unsigned char* buf, size_t total, int ld, int wgs, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(wk<MODE>, dim3(wgs), dim3(256), 0, 0, buf, total, ld);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  printf("%-34s total %4zu MB ld %5d WGs %5d: %6.0f GB/s (%.1f us)\n", name, total >> 20, ld, wgs, total / (best * 1e-3) / 1e9, best * 1e3);
}
int main() {
  unsigned char *buf, *src;
  const size_t maxb = (size_t)1 << 30;
  hipMalloc(&buf, maxb); hipMalloc(&src, maxb); hipMemset(buf, 0, maxb); hipMemset(src, 1, maxb);
  for (size_t total : {(size_t)48 << 20, (size_t)192 << 20, (size_t)768 << 20}) {
    for (int wgs : {256, 1024, 4096}) {
      run<0>(buf, total, 1536, wgs, "linear 16B");
      run<1>(buf, total, 1536, wgs, "gemm bf16 16B@32B x2, ld 1536");
      run<1>(buf, total, 512, wgs, "gemm bf16 16B@32B x2, ld 512");
      run<2>(buf, total, 1024, wgs, "gemm f32 16B@64B x4, ld 1024");
      run<2>(buf, total, 6144, wgs, "gemm f32 16B@64B x4, ld 6144");
      run<4>(buf, total, 1536, wgs, "full rows 512B, ld 1536");
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(copyk, dim3(4096), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)buf, total / 16);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("copy %zu MB: %.0f GB/s written (+ same read) (%.1f us)\n", total >> 20, total / (ms * 1e-3) / 1e9, ms * 1e3);
    }
  }
  return 0;
}
