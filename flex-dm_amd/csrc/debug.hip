// Hardware probe used by tests/test_gpu_kernels.py: pins the lane mapping of
// ds_read_b64_tr_b16 that gemm.hip / attention.hip rely on:
//   within each 16-lane group, out[lane i][j] = in[lane 4j + i/4][i % 4]
// where in[p][.] are the four b16 values at lane p's address.
#include "common.h"

__global__ void tr_probe_kernel(const int* __restrict__ byte_addr, unsigned short* __restrict__ outv) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[2048];
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned short* ptr = lds + byte_addr[threadIdx.x] / 2;
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((bf16x4 MFP_LDS*)ptr);
  for (int j = 0; j < 4; ++j) outv[threadIdx.x * 4 + j] = (unsigned short)v[j];
}

// byte_addr int32 [64] (8-byte aligned offsets into a 4096-byte LDS image whose b16 element e
// holds the value e); out uint16 [64][4].
extern "C" int mfp_debug_tr_probe(const int32_t* byte_addr, uint16_t* out, mfp_stream_t stream) {
  MFP_CHECK_ARG(byte_addr && out);
  hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), byte_addr, out);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}

// Hardware probe for v_mfma_scale_f32_16x16x128_f8f6f4 (tests/test_gpu_kernels.py pins the operand / scale layout
// gemm_fp8.hip relies on): a, b uint8 [64 lanes][32] (e4m3), sa, sb int32 [64] (scale VGPR per lane, byte 0 used),
// out f32 [64 lanes][4].
typedef __attribute__((ext_vector_type(8))) int dbg_i32x8;
__global__ void mx_probe_kernel(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                const int* __restrict__ sa, const int* __restrict__ sb, float* __restrict__ out) {
  const int l = threadIdx.x;
  dbg_i32x8 av, bv;
  for (int i = 0; i < 8; ++i) {
    av[i] = reinterpret_cast<const int*>(a)[l * 8 + i];
    bv[i] = reinterpret_cast<const int*>(b)[l * 8 + i];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

extern "C" int mfp_debug_mx_probe(const uint8_t* a, const uint8_t* b, const int32_t* sa, const int32_t* sb, float* out,
                                  mfp_stream_t stream) {
  MFP_CHECK_ARG(a && b && sa && sb && out);
  hipLaunchKernelGGL(mx_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a, b, sa, sb, out);
  MFP_CHECK_LAUNCH();
  return MFP_OK;
}
