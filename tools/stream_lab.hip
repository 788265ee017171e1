// stream_lab.hip -- read bandwidth of a row-major A[m][K] (float) when each wave walks its 64 rows in blocks of
// R rows x C bytes (16 KiB per step, 16 x b128 per lane in flight), against a flat grid-stride read of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int R>   // R rows per step: 64, 16 or 4;  C = 16384 / R bytes per row per step
__global__ __launch_bounds__(256, 2) void k_blocks(const float* __restrict__ A, int64_t lda, int K, float* out) {
  constexpr int CB = 16384 / R;            // bytes of a row per step
  constexpr int LPR = CB / 16 < 64 ? CB / 16 : 64;   // lanes per row in one instruction
  constexpr int RPI = 64 / LPR;            // rows per instruction
  constexpr int IPR = CB / 16 / LPR;       // instructions per row group
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wv) * 64;
  f32x4 s = {0, 0, 0, 0};
  for (int rb = 0; rb < 64; rb += R)
    for (int cb = 0; cb < K * 4; cb += CB) {
      f32x4 v[16];
      int q = 0;
#pragma unroll
      for (int rr = 0; rr < R; rr += RPI)
#pragma unroll
        for (int ii = 0; ii < IPR; ++ii, ++q)
          v[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(A + (row0 + rb + rr + lane / LPR) * lda) + cb + (ii * LPR + lane % LPR) * 16);
#pragma unroll
      for (int j = 0; j < 16; ++j) s += v[j];
    }
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}

__global__ __launch_bounds__(256, 2) void k_flat(const f32x4* __restrict__ A, size_t n16, float* out) {
  f32x4 s = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 15 * stride < n16; i += 16 * stride) {
    f32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = A[i + j * stride];
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
  }
  for (; i < n16; i += stride) s += A[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[0] = 1.f;
}

int main(int argc, char** argv) {
  const int64_t m = argc > 1 ? atoll(argv[1]) : 262144;
  const int K = argc > 2 ? atoi(argv[2]) : 1024;
  float *A, *out;
  CK(hipMalloc(&A, m * K * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(A, 0, m * K * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double bytes = (double)m * K * 4;
#define RUN(label, launch)                                                                     \
  for (int it = 0; it < 5; ++it) {                                                             \
    CK(hipEventRecord(e0)); launch; CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());        \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                            \
    if (it >= 3) printf("%-28s %.3f ms  %.2f TB/s\n", label, ms, bytes / ms / 1e9);            \
  }
  const int wgs = (int)(m / 256);
  RUN("64 rows x 256 B per step", (k_blocks<64><<<wgs, 256>>>(A, K, K, out)))
  RUN("16 rows x 1 KiB per step", (k_blocks<16><<<wgs, 256>>>(A, K, K, out)))
  RUN("4 rows x 4 KiB per step", (k_blocks<4><<<wgs, 256>>>(A, K, K, out)))
  RUN("flat, 2048 WGs", (k_flat<<<2048, 256>>>((const f32x4*)A, (size_t)(m * K / 4), out)))
  RUN("flat, 512 WGs", (k_flat<<<512, 256>>>((const f32x4*)A, (size_t)(m * K / 4), out)))
  RUN("flat, 8192 WGs", (k_flat<<<8192, 256>>>((const f32x4*)A, (size_t)(m * K / 4), out)))
  return 0;
}
