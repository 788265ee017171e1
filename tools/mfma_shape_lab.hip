// mfma_shape_lab.hip -- does v_mfma_f32_32x32x2_f32 buy the one-pass kernel anything over v_mfma_f32_16x16x4_f32?  
//
// The phase-B stream of k_nmf_fused at 32 bases (P += W_b^T V_b), isolated: one wave per SIMD, the V image of a block in LDS
// (filled by LDS-DMA, one 1-KiB request per step as in the production kernel, double buffered here), per step ONE ds_read_b128
// of the V image feeding the step's MFMAs, the A operand (the new W rows) in registers.  Same bytes per MFMA cycle in both forms:
//   SHAPE 16: 16-row block x 256 columns: 16 steps x 8 v_mfma_f32_16x16x4_f32   (NT = 2 base tiles x 4 column tiles; 32 cycles each)
//   SHAPE 32: 32-row block x 128 columns: 16 steps x 4 v_mfma_f32_32x32x2_f32   (1 base tile x 4 column tiles;      64 cycles each)
// i.e. 4 096 MFMA cycles and 16 KiB of V per block and wave either way.  Arguments: MODE bits 1 = LDS reads, 2 = LDS-DMA, 4 = V from
// HBM (a fresh 16 KiB per block) instead of one L2-resident tile.  Prints cycles per block (s_memtime, mean over waves) against 4 096.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_shape_lab.hip -o build_ab/mfma_shape_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pymf_amd/csrc/pmf_dev.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GLDS16(gsrc, ldst)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc), \
                                   (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int SHAPE, int MODE>
__global__ __launch_bounds__(256, 1) void k_lab(const float* __restrict__ V, int nblk, float* __restrict__ out,
                                                unsigned long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // 4 waves x 2 buffers x 16 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* sv = smem + wv * (2 * 4096);
  const int gw = blockIdx.x * 4 + wv;
  const char* src = reinterpret_cast<const char*>(V) + ((MODE & 4) ? (size_t)gw * nblk * 16384 : (size_t)0);
  const unsigned loff = 16u * lane;                                // a request: 64 lanes x 16 B = 1 KiB, linear in LDS
  // A operand: the new W rows of the block (registers; any values)
  float wa[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) wa[q] = 0.001f * (float)((lane * 7 + q * 13) & 63);
  f32x4 P16[2][16];
  f32x16 P32[4];
  if (SHAPE == 16) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 16; ++b) P16[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 16; ++b) P32[a][b] = 0.f;
  }
  auto issue = [&](int blk, int q, int buf) {                      // request q (0..15) of block blk into buffer buf
    if (MODE & 2) GLDS16(src + ((MODE & 4) ? (size_t)blk * 16384 : (size_t)0) + q * 1024 + loff, sv + buf * 4096 + q * 256);
  };
#pragma unroll
  for (int q = 0; q < 16; ++q) issue(0, q, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!(MODE & 2)) {                                               // no DMA: fill the images once with something
    for (int e = lane; e < 8192; e += 64) sv[e] = 0.5f + 0.001f * (float)(e & 255);
  }
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  const int i16 = lane & 15, kq = lane >> 4, c32 = lane & 31, h = lane >> 5;
  for (int b = 0; b < nblk; ++b) {
    const float* img = sv + (b & 1) * 4096;
    const int nb = (b + 1) & 1;
    const int nxt = b + 1 < nblk ? b + 1 : b;
    f32x4 bf[2];
    auto rd = [&](int s) -> f32x4 {
      if (!(MODE & 1)) return f32x4{1.f, 2.f, 3.f, 4.f};
      if (SHAPE == 16) {                       // image [4 panels][16 rows][64]: row 4 kq + j of panel p, chunk i
        const int p = s >> 2, j = s & 3;
        return *reinterpret_cast<const f32x4*>(img + p * 1024 + (4 * kq + j) * 64 + 4 * i16);
      } else {                                 // image [32 rows][128]: row 8 bb + 4 h + j, chunk c
        const int bb = s >> 2, j = s & 3;
        return *reinterpret_cast<const f32x4*>(img + (8 * bb + 4 * h + j) * 128 + 4 * c32);
      }
    };
    bf[0] = rd(0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s + 1 < 16) bf[(s + 1) & 1] = rd(s + 1);
      const f32x4 v = bf[s & 1];
      if (SHAPE == 16) {
        const int p = s >> 2, j = s & 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) P16[mt][4 * p + nt] = mfma16(wa[4 * mt + j], v[nt], P16[mt][4 * p + nt]);
      } else {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) P32[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], v[nt], P32[nt], 0, 0, 0);
      }
      issue(nxt, s, nb);                        // one request per step, into the other buffer
      if (s + 1 < 16) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    // the next block's image has landed (a full block of distance)
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float acc = 0.f;
  if (SHAPE == 16) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 16; ++b) acc += P16[a][b][0] + P16[a][b][1] + P16[a][b][2] + P16[a][b][3];
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 16; ++b) acc += P32[a][b];
  }
  out[(size_t)gw * 64 + lane] = acc;
  if (lane == 0) cyc[gw] = t1 - t0;
}

template <int SHAPE, int MODE>
int run(const float* V, int nblk, float* out, unsigned long long* cyc) {
  const size_t smem = 4 * 2 * 16384;
  CK(hipFuncSetAttribute((const void*)&k_lab<SHAPE, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 200; ++it) k_lab<SHAPE, MODE><<<256, 256, smem>>>(V, nblk, out, cyc);     // clock ramp
  CK(hipEventRecord(e0));
  for (int it = 0; it < 50; ++it) k_lab<SHAPE, MODE><<<256, 256, smem>>>(V, nblk, out, cyc);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(1024); CK(hipMemcpy(h.data(), cyc, 1024 * 8, hipMemcpyDeviceToHost));
  double s = 0; for (auto x : h) s += (double)x;
  printf("  shape %2d reads %d dma %d hbm %d: %8.1f cycles per block (4096 of MFMA; +%.1f %%), %.2f us per launch\n", SHAPE, MODE & 1, (MODE >> 1) & 1,
         (MODE >> 2) & 1, s / 1024 / nblk, (s / 1024 / nblk / 4096.0 - 1.0) * 100.0, ms / 50 * 1e3);
  return 0;
}

int main(int argc, char** argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 64;
  float *V, *out; unsigned long long* cyc;
  const size_t vbytes = (size_t)1024 * nblk * 16384;
  CK(hipMalloc(&V, vbytes)); CK(hipMemset(V, 0x3c, vbytes));
  CK(hipMalloc(&out, 1024 * 64 * 4)); CK(hipMalloc(&cyc, 1024 * 8));
  printf("phase-B stream, %d blocks per wave, 256 workgroups x 4 waves (one per SIMD)\n", nblk);
  if (run<16, 0>(V, nblk, out, cyc) || run<32, 0>(V, nblk, out, cyc)) return 1;     // bare MFMA streams
  if (run<16, 1>(V, nblk, out, cyc) || run<32, 1>(V, nblk, out, cyc)) return 1;     // + LDS reads
  if (run<16, 3>(V, nblk, out, cyc) || run<32, 3>(V, nblk, out, cyc)) return 1;     // + LDS-DMA (L2-resident source)
  if (run<16, 7>(V, nblk, out, cyc) || run<32, 7>(V, nblk, out, cyc)) return 1;     // + the source streamed from HBM
  return 0;
}
