// Microbenchmark: the transposed-product inner loop of the activation-stationary kernels, alone: 8 waves, wave (rp, nh),
// x fragments in registers (32 rows x K = 256), one 32 KB weight chunk [64][512 B] resident in LDS (slot ^ (row & 15)),
// acc[2 nt][2 rt] += W x^T: 32 MFMAs + 16 ds_read_b128 per wave and "chunk".  Prints microseconds per chunk per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>      // 0: as the kernels (wf double-buffered by hand), 1: MFMAs only, 2: LDS reads only,
                         // 3: the same product as v_mfma_f32_32x32x16_bf16 (wave = one 32 x 32 tile, 16 MFMAs + 16 ds_read_b128), 4: those MFMAs only
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
  const int nh = wave >> 2;
  for (int i = tid; i < 8192; i += 512) reinterpret_cast<unsigned int*>(smem)[i] = 0x3c003c00u + i;
  __syncthreads();
  bf16x8 xf[2][8];
  for (int rt = 0; rt < 2; ++rt) for (int ks = 0; ks < 8; ++ks) for (int e = 0; e < 8; ++e) xf[rt][ks][e] = (short)(0x3c00 + lane + ks + rt + e);
  int xs[4];
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;
  const unsigned char* wa = smem + ((nh * 2) * 16 + li) * 512;
  f32x4 acc[2][2] = {};
  f32x16 acc32 = {};
  f32x4 junk = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      bf16x8 wf[2][2];
      for (int nt = 0; nt < 2; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[0]);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8)
          for (int nt = 0; nt < 2; ++nt) wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[(ks + 1) & 3] + ((ks + 1) >> 2) * 256);
        for (int nt = 0; nt < 2; ++nt)
          for (int rt = 0; rt < 2; ++rt) acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], xf[rt][ks], acc[nt][rt], 0, 0, 0);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        for (int nt = 0; nt < 2; ++nt)
          for (int rt = 0; rt < 2; ++rt) acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[nt][ks], xf[rt][ks], acc[nt][rt], 0, 0, 0);
    } else if (MODE == 3 || MODE == 4) {
      // wave (rp, nh): rows 32 rp .. + 31 (B operand: lane l = row l & 31, k = 16 ks + 8 (l >> 5) .. + 7, from xf) x chunk columns
      // 32 nh .. + 31 (A operand: lane l = weight row 32 nh + (l & 31), 16-byte slot 2 ks + (l >> 5) of its 512-byte image row)
      const unsigned char* wr = smem + (nh * 32 + (lane & 31)) * 512;
      const int sw = lane & 15, hi = lane >> 5;
      bf16x8 wf2[2];
      if (MODE == 3) wf2[0] = *reinterpret_cast<const bf16x8*>(wr + (((0 + hi) ^ sw) << 4));
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        if (MODE == 3 && ks + 1 < 16) wf2[(ks + 1) & 1] = *reinterpret_cast<const bf16x8*>(wr + ((((ks + 1) * 2 + hi) ^ sw) << 4));
        acc32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MODE == 3 ? wf2[ks & 1] : xf[1][ks & 7], xf[ks >> 3][ks & 7], acc32, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        for (int nt = 0; nt < 2; ++nt) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[ks & 3] + (ks >> 2) * 256);
          junk[0] += (float)w[0]; junk[1] += (float)w[7];
        }
    }
    asm volatile("" ::: "memory");
  }
  float s = junk[0] + junk[1];
  for (int r = 0; r < 16; ++r) s += acc32[r];
  for (int nt = 0; nt < 2; ++nt) for (int rt = 0; rt < 2; ++rt) for (int r = 0; r < 4; ++r) s += acc[nt][rt][r];
  if (s == 1234.5f) out[0] = s;
}
template <int MODE>
void run(float* out, const char* what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 32768, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-28s %.3f us per chunk-product (128 rows x 64 cols x K 256 per CU)\n", what, ms * 1e3 / iters);
  }
}
int main() {
  float* out; hipMalloc(&out, 4);
  run<0>(out, "LDS reads + MFMAs (kernel)"); run<1>(out, "MFMAs only"); run<2>(out, "LDS reads only");
  run<3>(out, "32x32x16: LDS reads + MFMAs"); run<4>(out, "32x32x16: MFMAs only");
  return 0;
}
