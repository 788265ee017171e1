// colgemm_grid_lab.hip -- round 6: k_colgemm_stream<4,true> + k_reduce_slabs at cfg3's shape for several totals of workgroups
// (the library's rule: ~1 024 = 256 row chunks x 4 column panels; fewer chunks = fewer, longer workgroups and smaller slabs).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/colgemm_grid_lab.hip -o tools/colgemm_grid_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "../pymf_amd/csrc/pmf_dev.h"
#include "../pymf_amd/csrc/pmf_tiled.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (float)(x >> 8) * (1.0f / 16777216.0f);
  }
}
int main() {
  const int64_t m = 262144; const int n = 1024; constexpr int NT = 4, KP = 64;
  float *V, *W, *slab, *PS; double* Gd;
  CK(hipMalloc(&V, (size_t)m * n * 4)); CK(hipMalloc(&W, (size_t)m * KP * 4)); CK(hipMalloc(&PS, (size_t)KP * (n + KP) * 4)); CK(hipMalloc(&Gd, KP * KP * 8));
  k_fill<<<4096, 256>>>(V, (size_t)m * n, 1u); k_fill<<<1024, 256>>>(W, (size_t)m * KP, 3u);
  CK(hipMalloc(&slab, (size_t)4096 * KP * (n + KP) * 4));
  const size_t csm = (size_t)2 * 64 * (16 * NT + 4) * sizeof(float);
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  const int64_t E = (int64_t)KP * (n + KP);
  for (int want : {128, 192, 256, 384, 512, 1024}) {          // row chunks (x 4 panels = workgroups)
    int rpc = (int)(((m / 16 + want - 1) / want + 3) / 4 * 4 * 16); const int nch = (int)((m + rpc - 1) / rpc);
    auto run_col = [&] { hipLaunchKernelGGL((k_colgemm_stream<NT, true>), dim3((unsigned)nch, 4u), dim3(256), csm, 0, V, (int64_t)n, n, W, (int64_t)KP, m, rpc, slab, (int64_t)n + KP, 0); };
    auto run_red = [&] { hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((E / 4 + 63) / 64)), dim3(1024), 0, 0, slab, nch, E, PS, Gd, n, KP, KP); };
    for (int w = 0; w < 30; ++w) { run_col(); run_red(); }
    float tc = 0, tr = 0;
    for (int r = 0; r < 40; ++r) { CK(hipEventRecord(e0)); run_col(); CK(hipEventRecord(e1)); run_red(); CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
      float a, b; CK(hipEventElapsedTime(&a, e0, e1)); CK(hipEventElapsedTime(&b, e1, e2)); tc += a; tr += b; }
    CK(hipEventRecord(e0)); for (int r = 0; r < 40; ++r) { run_col(); run_red(); } CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float hot; CK(hipEventElapsedTime(&hot, e0, e1));
    printf("row chunks %4d (x4 = %4d workgroups, %5d rows each, slabs %5.1f MB): colgemm %.1f us  reduce %.1f us  pair back to back %.1f us\n", nch, 4 * nch, rpc,
           (double)nch * E * 4 / 1e6, tc / 40 * 1e3, tr / 40 * 1e3, hot / 40 * 1e3);
  }
  return 0;
}
