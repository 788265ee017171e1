// rowgemm_lab.hip -- stand-alone bench of plain-product kernels C[m][64] = A[m][K] B[64][K]^T (NMFALS' V H^T at cfg3:
// m = 262 144, K = 1 024) for quick iteration on the kernel structure.  hipcc --offload-arch=gfx950 -O3 -std=c++17
// -mllvm -amdgpu-mfma-vgpr-form=1 tools/rowgemm_lab.hip -o tools/rowgemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pymf_amd/csrc/pmf_dev.h"
#include "../pymf_amd/csrc/pmf_tiled.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ unsigned long long g_acc[8][4];   // [mode][store+wait, barrier, issue, reads+MFMA]  (wave 0 of block 100)
#define GLDS16(gsrc, ldst) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc), (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)

// Variant W: A fragments from global into registers one panel ahead, B panel in LDS (double buffered).
template <int NT, int MODE>        // MODE 0: the kernel; 1: loads only (one add per fragment instead of the MFMAs); 2: MFMAs only (A read once)
__global__ __launch_bounds__(256, 2) void k_wide(const float* __restrict__ A, int64_t lda, int kdim, const float* __restrict__ B, int64_t ldb,
                                                 float* __restrict__ C, int64_t ldc, int ntiles64) {
  constexpr int KP = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float sbw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int npan = kdim >> 6;
  const int tile64 = blockIdx.x * 4 + wv;
  const bool act = tile64 < ntiles64;
  const float* Arow = A + ((int64_t)(act ? tile64 : 0) * 64 + i) * lda + 4 * kq;
  const float* Arow3 = A + ((int64_t)(act ? tile64 : 0) * 64 + kq) * lda + 4 * i;
  f32x4 acc[4][NT];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rb][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 fa0[4][4], fa1[4][4];
  auto load_a = [&](int p, f32x4 (&fa)[4][4]) {
#pragma unroll
    for (int rb = 0; rb < (MODE == 4 ? 2 : 4); ++rb)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        fa[rb][t] = MODE == 3 ? *reinterpret_cast<const f32x4*>(Arow3 + (int64_t)(16 * rb + 4 * t) * lda + 64 * p)      // 4 rows x 256 B per instruction
                              : *reinterpret_cast<const f32x4*>(Arow + (int64_t)(16 * rb) * lda + 64 * p + 16 * t);      // 16 rows x 64 B
  };
  constexpr int BCH = KP * 16 / 256;                  // 16-byte pieces of a B panel per thread
  f32x4 pbr[BCH];
  auto load_b = [&](int p) {                           // unconditional (kdim % 64 == 0): no branch for the waitcnt pass to trip over
#pragma unroll
    for (int q = 0; q < BCH; ++q) {
      const int id = tid + 256 * q;
      pbr[q] = *reinterpret_cast<const f32x4*>(B + (int64_t)(id >> 4) * ldb + 64 * p + 4 * (id & 15));
    }
  };
  auto store_b = [&](float* cb) {
#pragma unroll
    for (int q = 0; q < BCH; ++q) {
      const int id = tid + 256 * q;
      lds_write4(cb, id >> 4, id & 15, pbr[q]);
    }
  };
  auto panel = [&](int p, f32x4 (&fa)[4][4], f32x4 (&fan)[4][4]) {
    float* cb = sbw + (p & 1) * (KP * 64);
    unsigned long long t0, t1, t2, t3, t4;
    STAMP(t0);
    store_b(cb);
    STAMP(t1);
    __syncthreads();
    STAMP(t2);
    const int pn = p + 1 < npan ? p + 1 : p;           // (the last panel re-requests itself: harmless, keeps the code straight-line)
    load_b(pn);
    if (MODE != 2 && MODE != 5) load_a(pn, fan);
    STAMP(t3);
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[rb][t] += fa[rb][t];
      return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 b4[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b4[nt] = lds_read4(cb, 16 * nt + i, 4 * t + kq);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        if (MODE == 5) {                                 // one request per 16 MFMAs instead of a burst of 16
          fan[rb][t] = *reinterpret_cast<const f32x4*>(Arow + (int64_t)(16 * rb) * lda + 64 * pn + 16 * t);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[rb][nt] = mfma16((MODE == 2 ? fa0 : fa)[MODE == 4 ? (rb & 1) : rb][t][e], b4[nt][e], acc[rb][nt]);
        if (MODE == 5) __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(t4);
    if (blockIdx.x == 100 && tid == 0) { g_acc[MODE][0] += t1 - t0; g_acc[MODE][1] += t2 - t1; g_acc[MODE][2] += t3 - t2; g_acc[MODE][3] += t4 - t3; }
  };
  load_b(0);
  load_a(0, fa0);
  for (int p = 0; p < npan; p += 2) {
    panel(p, fa0, fa1);
    panel(p + 1, fa1, fa0);                            // npan even (lab): a branch here lets LLVM sink the prefetch into it
  }
  if (act) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) C[((int64_t)tile64 * 64 + 16 * rb + 4 * kq + j) * ldc + 16 * nt + i] = acc[rb][nt][j];
  }
}

// Variant Q: as k_wide, but the A fragments travel in four register stages of half a panel (32 columns) each, requested
// three half panels ahead (1.5 panel times instead of 1, and the requests are spread out instead of one burst per panel).
template <int NT>
__global__ __launch_bounds__(256, 2) void k_quarter(const float* __restrict__ A, int64_t lda, int kdim, const float* __restrict__ B, int64_t ldb,
                                                    float* __restrict__ C, int64_t ldc, int ntiles64) {
  constexpr int KP = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float sbw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int nhalf = kdim >> 5;
  const int tile64 = blockIdx.x * 4 + wv;
  const bool act = tile64 < ntiles64;
  const float* Arow = A + ((int64_t)(act ? tile64 : 0) * 64 + i) * lda + 4 * kq;
  f32x4 acc[4][NT];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rb][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 h0[4][2], h1[4][2], h2[4][2], h3[4][2];
  auto load_a = [&](int q, f32x4 (&h)[4][2]) {
    const int qq = q < nhalf ? q : nhalf - 1;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int t = 0; t < 2; ++t) h[rb][t] = *reinterpret_cast<const f32x4*>(Arow + (int64_t)(16 * rb) * lda + 32 * qq + 16 * t);
  };
  constexpr int BCH = KP * 16 / 256;
  f32x4 pbr[BCH];
  auto load_b = [&](int p) {
    const int pp = 2 * p < nhalf ? p : (nhalf >> 1) - 1;
#pragma unroll
    for (int q = 0; q < BCH; ++q) {
      const int id = tid + 256 * q;
      pbr[q] = *reinterpret_cast<const f32x4*>(B + (int64_t)(id >> 4) * ldb + 64 * pp + 4 * (id & 15));
    }
  };
  auto store_b = [&](float* cb) {
#pragma unroll
    for (int q = 0; q < BCH; ++q) {
      const int id = tid + 256 * q;
      lds_write4(cb, id >> 4, id & 15, pbr[q]);
    }
  };
  auto half = [&](const float* cb, int th, f32x4 (&h)[4][2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 b4[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b4[nt] = lds_read4(cb, 16 * nt + i, 4 * (2 * th + t) + kq);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[rb][nt] = mfma16(h[rb][t][e], b4[nt][e], acc[rb][nt]);
    }
  };
  load_b(0);
  load_a(0, h0); load_a(1, h1); load_a(2, h2);
  store_b(sbw);
  load_b(1);
  for (int q = 0; q < nhalf; q += 4) {                 // nhalf % 4 == 0 (lab)
    // The B panel for the next barrier is stored at the END of the previous stretch, so that its wait sits where the
    // count of younger requests is known exactly (at the loop head the compiler falls back to vmcnt(0)).
    __syncthreads();
    load_a(q + 3, h3);
    __builtin_amdgcn_sched_barrier(0);
    half(sbw, 0, h0);
    load_a(q + 4, h0);
    __builtin_amdgcn_sched_barrier(0);
    half(sbw, 1, h1);
    store_b(sbw + KP * 64);
    load_b((q >> 1) + 2);
    __syncthreads();
    load_a(q + 5, h1);
    __builtin_amdgcn_sched_barrier(0);
    half(sbw + KP * 64, 0, h2);
    load_a(q + 6, h2);
    __builtin_amdgcn_sched_barrier(0);
    half(sbw + KP * 64, 1, h3);
    store_b(sbw);
    load_b((q >> 1) + 3);
  }
  if (act) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) C[((int64_t)tile64 * 64 + 16 * rb + 4 * kq + j) * ldc + 16 * nt + i] = acc[rb][nt][j];
  }
}

// Variant I: A fragments from global into registers one panel ahead with the requests INTERLEAVED with the MFMAs
// (one b128 request per 16 MFMAs): a burst of 20 requests blocks the wave at issue for as long as the panel's MFMAs take.
template <int NT>
__global__ __launch_bounds__(256, 2) void k_inter(const float* __restrict__ A, int64_t lda, int kdim, const float* __restrict__ B, int64_t ldb,
                                                  float* __restrict__ C, int64_t ldc, int ntiles64) {
  constexpr int KP = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float sbw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int npan = kdim >> 6;
  const int tile64 = blockIdx.x * 4 + wv;
  const bool act = tile64 < ntiles64;
  const float* Arow = A + ((int64_t)(act ? tile64 : 0) * 64 + i) * lda + 4 * kq;
  f32x4 acc[4][NT];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[rb][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 fa0[4][4], fa1[4][4];
  constexpr int BCH = KP * 16 / 256;
  f32x4 pbr[BCH];
  auto load_b = [&](int p) {
    const int pp = p < npan ? p : npan - 1;
#pragma unroll
    for (int q = 0; q < BCH; ++q) {
      const int id = tid + 256 * q;
      pbr[q] = *reinterpret_cast<const f32x4*>(B + (int64_t)(id >> 4) * ldb + 64 * pp + 4 * (id & 15));
    }
  };
  auto store_b = [&](float* cb) {
#pragma unroll
    for (int q = 0; q < BCH; ++q) {
      const int id = tid + 256 * q;
      lds_write4(cb, id >> 4, id & 15, pbr[q]);
    }
  };
  auto panel = [&](int p, f32x4 (&fa)[4][4], f32x4 (&fan)[4][4]) {
    float* cb = sbw + (p & 1) * (KP * 64);
    const int pn = p + 1 < npan ? p + 1 : p;
    const float* An = Arow + 64 * pn;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 b4[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b4[nt] = lds_read4(cb, 16 * nt + i, 4 * t + kq);
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        fan[rb][t] = *reinterpret_cast<const f32x4*>(An + (int64_t)(16 * rb) * lda + 16 * t);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[rb][nt] = mfma16(fa[rb][t][e], b4[nt][e], acc[rb][nt]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    store_b(sbw + ((p + 1) & 1) * (KP * 64));          // panel p + 1 of B: its buffer was last read in panel p - 1
    load_b(p + 2);
  };
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int t = 0; t < 4; ++t) fa0[rb][t] = *reinterpret_cast<const f32x4*>(Arow + (int64_t)(16 * rb) * lda + 16 * t);
  load_b(0);
  store_b(sbw);
  load_b(1);
  for (int p = 0; p < npan; p += 2) {                  // npan even (lab)
    panel(p, fa0, fa1);
    panel(p + 1, fa1, fa0);
  }
  if (act) {
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) C[((int64_t)tile64 * 64 + 16 * rb + 4 * kq + j) * ldc + 16 * nt + i] = acc[rb][nt][j];
  }
}

__global__ void fillk(float* p, size_t n, unsigned seed) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = u01_from(seed, i); }

int main(int argc, char** argv) {
  const int64_t m = argc > 1 ? atoll(argv[1]) : 262144;
  const int K = argc > 2 ? atoi(argv[2]) : 1024;
  constexpr int NT = 4, KP = 64;
  float *A, *B, *C, *C2;
  CK(hipMalloc(&A, m * K * 4)); CK(hipMalloc(&B, (size_t)KP * K * 4)); CK(hipMalloc(&C, m * KP * 4)); CK(hipMalloc(&C2, m * KP * 4));
  fillk<<<(m * K + 255) / 256, 256>>>(A, m * K, 1); fillk<<<(KP * K + 255) / 256, 256>>>(B, (size_t)KP * K, 2);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * m * K * KP;
  // reference: the library's k_rowgemm<4, EPI_STORE>
  {
    const size_t smem = rowgemm_smem_bytes<NT>();
    CK(hipFuncSetAttribute((const void*)&k_rowgemm<NT, EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    const int ntiles = (int)(m / 64), tpw = ntiles >= 8192 ? 8 : ntiles >= 2048 ? 4 : ntiles >= 1024 ? 2 : 1;
    for (int it = 0; it < 6; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_rowgemm<NT, EPI_STORE>), dim3((ntiles + tpw - 1) / tpw), dim3(256), smem, 0, A, (int64_t)K, K, B, (int64_t)K, (float*)nullptr, (const float*)nullptr, C, (int64_t)KP, 0.f, m, KP, ntiles, tpw);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 3) printf("k_rowgemm<4,store>: %.3f ms  %.1f TFLOP/s\n", ms, flop / ms / 1e9);
    }
  }
  const size_t smem = 2 * KP * 64 * 4;
  const int ntiles64 = (int)(m / 64);
#define RUNW(MODE, OUT)                                                                                                             \
  for (int it = 0; it < 5; ++it) {                                                                                                  \
    CK(hipEventRecord(e0));                                                                                                         \
    hipLaunchKernelGGL((k_wide<NT, MODE>), dim3((ntiles64 + 3) / 4), dim3(256), smem, 0, A, (int64_t)K, K, B, (int64_t)K, OUT, (int64_t)KP, ntiles64); \
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());                                                                             \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                                                                 \
    if (it >= 3) printf("k_wide<4,mode %d>:   %.3f ms  %.1f TFLOP/s  %.2f TB/s of A\n", MODE, ms, flop / ms / 1e9, m * K * 4.0 / ms / 1e9); \
  }
  RUNW(1, C) RUNW(2, C) RUNW(4, C) RUNW(0, C2) RUNW(5, C2)
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_quarter<NT>), dim3((ntiles64 + 3) / 4), dim3(256), smem, 0, A, (int64_t)K, K, B, (int64_t)K, C2, (int64_t)KP, ntiles64);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 3) printf("k_quarter<4>:        %.3f ms  %.1f TFLOP/s\n", ms, flop / ms / 1e9);
  }
  {
    const size_t smem2 = rowgemm_smem_bytes<NT>();
    const int ntiles = (int)(m / 64), tpw = ntiles >= 8192 ? 8 : ntiles >= 2048 ? 4 : ntiles >= 1024 ? 2 : 1;
    hipLaunchKernelGGL((k_rowgemm<NT, EPI_STORE>), dim3((ntiles + tpw - 1) / tpw), dim3(256), smem2, 0, A, (int64_t)K, K, B, (int64_t)K, (float*)nullptr, (const float*)nullptr, C, (int64_t)KP, 0.f, m, KP, ntiles, tpw);
    CK(hipDeviceSynchronize());
  }
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_inter<NT>), dim3((ntiles64 + 3) / 4), dim3(256), smem, 0, A, (int64_t)K, K, B, (int64_t)K, C2, (int64_t)KP, ntiles64);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 3) printf("k_inter<4>:          %.3f ms  %.1f TFLOP/s\n", ms, flop / ms / 1e9);
  }
  for (int it = 0; it < 6; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rowgemm_stream<NT, 4, EPI_STORE>), dim3(512), dim3(256), smem, 0, A, (int64_t)K, K, B, (int64_t)K, (float*)nullptr, (const float*)nullptr, C2, (int64_t)KP, 0.f, m, KP, ntiles64, (int64_t)KP);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 3) printf("k_rowgemm_stream<4,4>: %.3f ms  %.1f TFLOP/s\n", ms, flop / ms / 1e9);
  }
  for (int it = 0; it < 6; ++it) {       // 32-row wave tiles: half the registers, twice the waves
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rowgemm_stream<NT, 2, EPI_STORE>), dim3(512), dim3(256), smem, 0, A, (int64_t)K, K, B, (int64_t)K, (float*)nullptr, (const float*)nullptr, C2, (int64_t)KP, 0.f, m, KP, (int)(m / 32), (int64_t)KP);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (it >= 3) printf("k_rowgemm_stream<4,2>: %.3f ms  %.1f TFLOP/s\n", ms, flop / ms / 1e9);
  }
  {
    unsigned long long ha[8][4];
    CK(hipMemcpyFromSymbol(ha, HIP_SYMBOL(g_acc), sizeof(ha)));
    for (int md : {2, 4, 0, 5}) printf("mode %d stamps per panel (100 MHz ticks x10 = ns): store+wait %.0f  barrier %.0f  issue %.0f  reads+MFMA %.0f\n", md,
                                    ha[md][0] * 10.0 / (5 * (K / 64)), ha[md][1] * 10.0 / (5 * (K / 64)), ha[md][2] * 10.0 / (5 * (K / 64)), ha[md][3] * 10.0 / (5 * (K / 64)));
  }
  std::vector<float> h1(64 * KP), h2(64 * KP);
  CK(hipMemcpy(h1.data(), C + (m - 64) * KP, h1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), C2 + (m - 64) * KP, h2.size() * 4, hipMemcpyDeviceToHost));
  double md = 0; for (size_t q = 0; q < h1.size(); ++q) md = fmax(md, fabs(h1[q] - h2[q]) / fmax(1.0, fabs(h1[q])));
  printf("max rel diff (last 64 rows) %.2e\n", md);
  return 0;
}
#define INST(NT_, RB_, EPI_) template __global__ void k_rowgemm_stream<NT_, RB_, EPI_>(const float*, int64_t, int, const float*, int64_t, float*, const float*, float*, int64_t, float, int64_t, int, int, int64_t);
INST(1, 4, EPI_STORE) INST(2, 4, EPI_STORE) INST(8, 2, EPI_STORE)
INST(1, 4, EPI_NMF_W) INST(2, 4, EPI_NMF_W) INST(4, 4, EPI_NMF_W) INST(8, 2, EPI_NMF_W) INST(4, 4, EPI_BNMF_W) INST(8, 2, EPI_RNMF_W)
template __global__ void k_rowgemm_stream<8, 2, EPI_NMF_W, true>(const float*, int64_t, int, const float*, int64_t, float*, const float*, float*, int64_t, float, int64_t, int, int, int64_t);
