// gemm_ab.hip -- the PRODUCTION stream kernels of pmf_tiled.h at cfg3's shape (262 144 x 1 024, k = 64), timed and
// checksummed, for quick A/B of edits to the kernels themselves (compiles in seconds):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/gemm_ab.hip -o tools/gemm_ab && tools/gemm_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "../pymf_amd/csrc/pmf_dev.h"
#include "../pymf_amd/csrc/pmf_tiled.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (float)(x >> 8) * (1.0f / 16777216.0f);
  }
}
int main(int argc, char** argv) {
  const int64_t m = argc > 1 ? atoll(argv[1]) : 262144; const int n = argc > 2 ? atoi(argv[2]) : 1024; constexpr int NT = 4, KP = 64, RB = 4;
  float *V, *H, *W, *C, *slab;
  CK(hipMalloc(&V, (size_t)m * n * 4)); CK(hipMalloc(&H, (size_t)KP * n * 4)); CK(hipMalloc(&W, (size_t)m * KP * 4)); CK(hipMalloc(&C, (size_t)m * KP * 4));
  k_fill<<<4096, 256>>>(V, (size_t)m * n, 1u); k_fill<<<256, 256>>>(H, (size_t)KP * n, 2u); k_fill<<<1024, 256>>>(W, (size_t)m * KP, 3u);
  // ---- k_rowgemm_stream<4,4,STORE>: C = V H^T ----
  const int ntiles = (int)(m / (16 * RB)), ngroups = (ntiles + 3) / 4;
  const unsigned grid = (unsigned)std::min(ngroups, 512);
  const size_t ssm = rowgemm_stream_smem_bytes<NT, EPI_STORE, false>();
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run_row = [&] { hipLaunchKernelGGL((k_rowgemm_stream<NT, RB, EPI_STORE>), dim3(grid), dim3(256), ssm, 0, V, (int64_t)n, n, H, (int64_t)n, (float*)nullptr, (const float*)nullptr, C,
                                          (int64_t)KP, 0.f, m, KP, ntiles, (int64_t)KP); };
  // ---- k_colgemm_stream<4,true>: slabs of (W^T V | W^T W) ----
  const int n_panels = (n + 255) / 256; int nch = std::max(1, 1024 / n_panels);
  int rpc = (int)(((m / 16 + nch - 1) / nch + 3) / 4 * 4 * 16); nch = (int)((m + rpc - 1) / rpc);
  CK(hipMalloc(&slab, (size_t)nch * KP * (n + KP) * 4));
  const size_t csm = (size_t)2 * 64 * (16 * NT + 4) * sizeof(float);
  auto run_col = [&] { hipLaunchKernelGGL((k_colgemm_stream<NT, true>), dim3((unsigned)nch, (unsigned)n_panels), dim3(256), csm, 0, V, (int64_t)n, n, W, (int64_t)KP, m, rpc, slab,
                                          (int64_t)n + KP, 0); };
  for (int which = 0; which < 2; ++which) {
    auto run = [&] { if (which == 0) run_row(); else run_col(); };
    for (int w = 0; w < 20; ++w) run();
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < 20; ++r) { CK(hipEventRecord(e0)); run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms; }
    const double fl = which == 0 ? 2.0 * m * n * KP : 2.0 * m * n * KP + 2.0 * m * KP * KP;
    printf("%s  mean %.4f ms  best %.4f ms  %.1f TFLOP/s (mean)\n", which == 0 ? "k_rowgemm_stream<4,4,store>" : "k_colgemm_stream<4,true>   ", sum / 20, best, fl / (sum / 20 * 1e-3) / 1e12);
  }
  // checksums (order-sensitive): any change of the summation order shows
  std::vector<float> hc((size_t)1 << 20), hs((size_t)KP * (n + KP));
  CK(hipMemcpy(hc.data(), C + ((size_t)m * KP / 2), hc.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), slab + (size_t)(nch / 2) * KP * (n + KP), hs.size() * 4, hipMemcpyDeviceToHost));
  unsigned long long hcs = 1469598103934665603ull, hss = hcs;
  for (float v : hc) { unsigned u; memcpy(&u, &v, 4); hcs = (hcs ^ u) * 1099511628211ull; }
  for (float v : hs) { unsigned u; memcpy(&u, &v, 4); hss = (hss ^ u) * 1099511628211ull; }
  printf("checksums: C %016llx  slab %016llx\n", hcs, hss);
  return 0;
}
