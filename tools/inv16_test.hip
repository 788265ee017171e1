// diagnostic: inv16_wave (pmf_inv.h) on one random SPD 16 x 16 tile with unit diagonal
#include <hip/hip_runtime.h>
#include "/root/repo/pymf_amd/csrc/pmf_inv.h"
#include <cstdio>
#include <cmath>
#include <cstdlib>
__global__ void k(const double* A, double* out) {
  __shared__ double src[256], dst[256];
  __shared__ __attribute__((aligned(32))) double line[64];
  const int lane = threadIdx.x, g = lane >> 4, cc = lane & 15;
  for (int r = 0; r < 4; ++r) src[tile_lds_index(g + 4 * r, cc)] = A[(g + 4 * r) * 16 + cc];
  __syncthreads();
  inv16_wave(src, dst, line, lane);
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
  double *dA, *dI; hipMalloc(&dA, 2048); hipMalloc(&dI, 2048);
  hipMemcpy(dA, A, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dI);
  hipMemcpy(I, dI, 2048, hipMemcpyDeviceToHost);
  double worst = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int t = 0; t < 16; ++t) s += A[i * 16 + t] * I[t * 16 + j]; double e = fabs(s - (i == j)); if (!(e <= worst)) worst = e; }
  printf("inv16_wave: max |A inv - I| = %.3e  (inv[0][0]=%g inv[3][7]=%g inv[7][3]=%g)\n", worst, I[0], I[3*16+7], I[7*16+3]);
  return 0;
}
