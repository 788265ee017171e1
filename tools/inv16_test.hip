// diagnostic: inv16_wave (pmf_inv.h) on one random SPD 16 x 16 tile with unit diagonal
#include <hip/hip_runtime.h>
#ifndef PMF_INV_HEADER
#define PMF_INV_HEADER "/root/repo/pymf_amd/csrc/pmf_inv.h"
#endif
#include PMF_INV_HEADER
#ifndef INV16_PAIR
#define INV16_PAIR true
#endif
#include <cstdio>
#include <cmath>
#include <cstdlib>
__global__ void k(const double* A, double* out, unsigned long long* cyc) {
  __shared__ double src[256], dst[256];
  __shared__ __attribute__((aligned(32))) double2 line[128];
  const int lane = threadIdx.x, g = lane >> 4, cc = lane & 15;
  for (int r = 0; r < 4; ++r) src[tile_lds_index(g + 4 * r, cc)] = A[(g + 4 * r) * 16 + cc];
  __syncthreads();
  inv16_wave<INV16_PAIR>(src, dst, line, lane);
  __syncthreads();
  unsigned long long t0, t1;                  // cycles of one in-wave inverse, 64 in a row (s_memtime: 100 MHz ... wall_clock; use clock64)
  const unsigned long long w0 = wall_clock64();
  t0 = clock64();
  for (int rep = 0; rep < 64; ++rep) { inv16_wave<INV16_PAIR>(src, dst, line, lane); __builtin_amdgcn_s_waitcnt(0); }
  t1 = clock64();
  if (lane == 0) { cyc[0] = (t1 - t0) / 64; cyc[1] = t1 - t0; cyc[2] = wall_clock64() - w0; }
  __syncthreads();
  for (int r = 0; r < 4; ++r) out[(g + 4 * r) * 16 + cc] = dst[tile_lds_index(g + 4 * r, cc)];
}
int main() {
  double H[16][40], A[256], I[256];
  srand(3);
  for (auto& row : H) for (auto& x : row) x = rand() / (double)RAND_MAX;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int t = 0; t < 40; ++t) s += H[i][t] * H[j][t]; A[i * 16 + j] = s; }
  double sc[16]; for (int i = 0; i < 16; ++i) sc[i] = 1 / sqrt(A[i * 16 + i]);
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] *= sc[i] * sc[j];
  double *dA, *dI; unsigned long long *dC, hC = 0, hC3[3] = {0, 0, 0}; hipMalloc(&dA, 2048); hipMalloc(&dI, 2048); hipMalloc(&dC, 24);
  hipMemcpy(dA, A, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dI, dC);
  hipMemcpy(I, dI, 2048, hipMemcpyDeviceToHost); hipMemcpy(hC3, dC, 24, hipMemcpyDeviceToHost); hC = hC3[0];
  printf("shader clock while one wave runs alone: %.0f MHz (%llu cycles in %llu ticks of 10 ns)\n", hC3[2] ? 100.0 * hC3[1] / hC3[2] : 0.0, hC3[1], hC3[2]);
  double worst = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int t = 0; t < 16; ++t) s += A[i * 16 + t] * I[t * 16 + j]; double e = fabs(s - (i == j)); if (!(e <= worst)) worst = e; }
  printf("inv16_wave: %llu cycles per tile; max |A inv - I| = %.3e  (inv[0][0]=%g inv[3][7]=%g inv[7][3]=%g)\n", hC, worst, I[0], I[3*16+7], I[7*16+3]);
  return 0;
}
