// Microbenchmark: the weight stream of the activation-stationary kernels, alone.  Every workgroup (8 waves, one per CU)
// pulls the SAME 1 MB of weights through LDS in 32 KB chunks by LDS-DMA (each wave 4 x 1 KB per chunk), DEPTH chunks in
// flight, counted waits + one barrier per chunk, no compute.  Prints microseconds per chunk.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) unsigned char lds_u8;
template <int DEPTH>
__global__ __launch_bounds__(512) void k(const unsigned short* __restrict__ W, int nchunks, int reps, unsigned int* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, (unsigned int)nchunks * 32768u, 0x00020000);
  auto wload = [&](int c) {
    unsigned char* dst = smem + (c % DEPTH) * 32768 + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_u8*)(dst + i * 1024), 16, (unsigned int)(wave * 4096 + i * 1024 + lane * 16), (c % nchunks) * 32768, 0, 0);
  };
  const int total = nchunks * reps;
  for (int c = 0; c < DEPTH - 1; ++c) wload(c);
  unsigned int acc = 0;
  for (int c = 0; c < total; ++c) {
    if (c + DEPTH - 1 < total) wload(c + DEPTH - 1);
    // chunk c landed: at most the (DEPTH - 1) younger chunks' 4 loads each are outstanding
    if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc += *reinterpret_cast<const unsigned int*>(smem + (c % DEPTH) * 32768 + threadIdx.x * 4);
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 0x12345678u) out[0] = 1;
}
template <int DEPTH>
void run(const unsigned short* W, unsigned int* out, int wgs) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 32768);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nchunks = 32, reps = 8;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<DEPTH>, dim3(wgs), dim3(512), DEPTH * 32768, 0, W, nchunks, reps, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("WGs %4d depth %d: %.3f us per 32 KB chunk  (%.1f TB/s from L2 over the chip)\n", wgs, DEPTH, ms * 1e3 / (nchunks * reps),
                    (double)wgs * nchunks * reps * 32768 / (ms * 1e-3) / 1e12);
  }
}
int main() {
  unsigned short* W; unsigned int* out;
  hipMalloc(&W, 1 << 20); hipMemset(W, 1, 1 << 20); hipMalloc(&out, 4);
  for (int wgs : {256, 128, 32}) { run<2>(W, out, wgs); run<3>(W, out, wgs); run<4>(W, out, wgs); }
  return 0;
}
