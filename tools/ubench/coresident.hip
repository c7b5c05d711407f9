// Microbenchmark (round 6, VERDICT r05 #1 option (a)): the weight-chunk loop of the activation-stationary block kernels -- LDS-DMA of
// the chunk two ahead, fragment reads + products, bias / pack epilogue into a swizzled image, stash of the previous image to HBM,
// counted wait, barrier (tools/ubench/pingpong.hip MODE 0) -- as
//   A  ONE eight-wave workgroup per CU owning 128 rows, 64-column chunks (32 KB), 160 KB-class LDS: the shipped tile;
//   B  TWO independent four-wave workgroups per CU owning 64 rows each, 32-column chunks (16 KB) through a 3 x 16 KB ring + three
//      4 KB images = 60 KB of LDS per workgroup: what "two co-resident workgroups per CU at <= 80 KB" means for this loop.  Nothing
//      synchronises the two: they drift apart and one's products can run under the other's epilogue / DMA issue / stash;
//   C  the same four-wave workgroup ALONE on its CU (256 workgroups): what one of B's workgroups costs without a neighbour;
//   D  ONE eight-wave workgroup per CU owning 64 rows (one 16-row tile per wave), 64-column chunks: the half-tile form of round 5.
// Prints microseconds per 128 rows x 64 columns x K 256 of products per CU (the unit of pingpong.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
__device__ __forceinline__ unsigned int pk(float a, float b) { const f2 v = {a, b}; return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf2)); }

// ROWS rows per workgroup, CC columns per chunk, NW waves; wave (rp, nh): RT row tiles x NT column tiles of 16
template <int ROWS, int CC, int NW>
struct Cfg {
  static constexpr int NRP = (ROWS == 128 || NW == 4) ? ROWS / 32 : ROWS / 16;      // row groups of waves (NW 16: 4 x 4 waves, one column tile each)
  static constexpr int RT = ROWS / (16 * NRP);                                       // row tiles per wave
  static constexpr int NNH = NW / NRP;                                               // column groups of waves
  static constexpr int NT = CC / (16 * NNH);                                         // column tiles per wave
  static constexpr int WSB = CC * 512;                                               // bytes of a weight chunk [CC][512 B]
  static constexpr int IMGB = ROWS * CC * 2;                                         // image [ROWS][CC bf16]
  static constexpr int P = WSB / 1024 / NW;                                          // LDS-DMA pieces per wave and chunk
  static constexpr int S = IMGB / 16 / (NW * 64);                                    // stash stores per thread and chunk
  static constexpr int LDS = 3 * IMGB + 3 * WSB + 1024;
};

template <int ROWS, int CC, int NW, int OCC>
__global__ __launch_bounds__(NW * 64, OCC) void kc(const unsigned short* W, unsigned short* out, int nch, int nwbytes, unsigned int obytes) {
  using C = Cfg<ROWS, CC, NW>;
  constexpr int RT = C::RT, NT = C::NT, NRP = C::NRP, P = C::P, S = C::S, NTH = NW * 64;
  constexpr int SLOTS = CC * 2 / 16;      // 16-byte slots of an image row
  static_assert(P >= 1 && S >= 1 && RT >= 1 && NT >= 1, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const Im = smem;
  unsigned char* const Ws = smem + 3 * C::IMGB;
  float* const Bq = reinterpret_cast<float*>(smem + 3 * C::IMGB + 3 * C::WSB);
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wave % NRP, nh = wave / NRP;
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(W), 0, nwbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(out, 0, obytes, 0x00020000);
  if (tid < 256) Bq[tid] = 0.01f * tid;
  bf16x8 xf[RT][8];
  for (int rt = 0; rt < RT; ++rt) for (int ks = 0; ks < 8; ++ks) for (int e = 0; e < 8; ++e) xf[rt][ks][e] = (short)(0x3c00 + ((lane * 7 + ks * 3 + rt + e) & 63));
  int xs[4];
  for (int ks = 0; ks < 4; ++ks) xs[ks] = ((ks * 4 + g) ^ li) << 4;
  const int row0 = blockIdx.x * ROWS;
  auto isw = [](int row) { return (row >> 1) & (SLOTS - 1); };

  auto wload = [&](int c) {      // chunk rows wave * 2 P + 2 i + (lane >> 5), 16-byte slot ^ (row & 15)
    unsigned char* dst = Ws + (c % 3) * C::WSB + wave * (P * 1024);
    const int base = (int)(((long long)c * C::WSB) % nwbytes);
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int row = wave * 2 * P + 2 * i + (lane >> 5);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_u8*)(dst + i * 1024), 16, (unsigned int)(row * 512 + (((lane & 31) ^ (row & 15)) << 4)), base, 0, 0);
    }
  };
  auto stash = [&](int c) {
    const unsigned char* img = Im + (c % 3) * C::IMGB;
#pragma unroll
    for (int i = 0; i < S; ++i) {
      const int idx = tid + NTH * i, r = idx / SLOTS, c16 = idx % SLOTS;
      const u32x4 v = *reinterpret_cast<const u32x4*>(img + r * (CC * 2) + ((c16 ^ isw(r)) << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, rs_o, (unsigned int)(row0 + r) * 1536 + (unsigned int)((c * CC * 2) % 1536) + c16 * 16, 0, 0);
    }
  };
  wload(0);
  wload(1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int c = 0; c < nch; ++c) {
    f32x4 acc[NT][RT];
    wload(c + 2);
    if (c >= 1) stash(c - 1);
    const unsigned char* wa = Ws + (c % 3) * C::WSB + ((nh * NT) * 16 + li) * 512;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[nt][rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[2][NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wf[0][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[0]);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(wa + nt * 8192 + xs[(ks + 1) & 3] + ((ks + 1) >> 2) * 256);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[nt][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks & 1][nt], xf[rt][ks], acc[nt][rt], 0, 0, 0);
    }
    unsigned char* img = Im + (c % 3) * C::IMGB;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(Bq + (c & 3) * 64 + ((nh * NT + nt) * 16 + 4 * g) % 64);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const int row = rp * (16 * RT) + rt * 16 + li;
        const u32x2 p2 = {pk(acc[nt][rt][0] + bb[0], acc[nt][rt][1] + bb[1]), pk(acc[nt][rt][2] + bb[2], acc[nt][rt][3] + bb[3])};
        *reinterpret_cast<u32x2*>(img + row * (CC * 2) + ((((nh * NT + nt) * 2 + (g >> 1)) ^ isw(row)) << 4) + (g & 1) * 8) = p2;
      }
    }
    // chunk c + 1 landed: younger than its loads are stash(c - 2) [S], wload(c + 2) [P], stash(c - 1) [S]
    if (c < 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P + 2 * S) : "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int ROWS, int CC, int NW, int OCC>
void run(const unsigned short* W, int nwbytes, unsigned short* out, unsigned int obytes, int grid, const char* what) {
  using C = Cfg<ROWS, CC, NW>;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // the same PRODUCTS per CU in every form: 480 units of 128 rows x 64 columns
  const int wg_per_cu = grid / 256;
  const long long unit = 128ll * 64, per_chunk = (long long)ROWS * CC * wg_per_cu;
  const int nch = (int)(480 * unit / per_chunk);
  auto fn = kc<ROWS, CC, NW, OCC>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, NW * 64, C::LDS);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(NW * 64), C::LDS, 0, W, out, nch, nwbytes, obytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("%-78s %.3f us per unit  (LDS %d B, %d wave(s), RT %d NT %d, %d DMA pieces + %d stash stores per wave-thread and chunk, occupancy %d / CU)\n",
                    what, ms * 1e3 / 480 * (256.0 * wg_per_cu / grid), C::LDS, NW, C::RT, C::NT, C::P, C::S, occ);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) printf("error: %s\n", hipGetErrorString(e));
}

int main() {
  unsigned short* W; unsigned short* out;
  const unsigned int obytes = 32768u * 1536u;
  const int nwbytes = 48 * 32768;
  hipMalloc(&W, nwbytes); hipMalloc(&out, obytes);
  unsigned short* h = (unsigned short*)malloc(nwbytes);
  for (int i = 0; i < nwbytes / 2; ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 26));
  hipMemcpy(W, h, nwbytes, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<128, 64, 8, 1>(W, nwbytes, out, obytes, 256, "A  one 8-wave workgroup per CU, 128 rows, 64-column chunks (the shipped tile)");
    run<64, 32, 4, 2>(W, nwbytes, out, obytes, 512, "B  TWO 4-wave workgroups per CU, 64 rows each, 32-column chunks, 60 KB LDS each");
    run<64, 32, 4, 2>(W, nwbytes, out, obytes, 256, "C  the same products by ONE such 4-wave workgroup per CU (no neighbour)");
    run<64, 64, 8, 1>(W, nwbytes, out, obytes, 256, "D  one 8-wave workgroup per CU, 64 rows, one row tile per wave (half tiles of r05), 64-column chunks");
    run<128, 64, 16, 1>(W, nwbytes, out, obytes, 256, "F  one SIXTEEN-wave workgroup per CU, 128 rows, 64-column chunks (4 waves per SIMD, <= 128 registers)");
    run<64, 64, 4, 2>(W, nwbytes, out, obytes, 512, "E  TWO 4-wave workgroups per CU, 64 rows, 64-column chunks (108 KB LDS: does not co-reside)");
  }
  return 0;
}
