// gemm_ab.hip -- the PRODUCTION stream kernels of pmf_tiled.h at cfg3's shape (262 144 x 1 024, k = 64), timed and
// checksummed, for quick A/B of edits to the kernels themselves (compiles in seconds):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/gemm_ab.hip -o tools/gemm_ab && tools/gemm_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <cmath>
#include <algorithm>
#include "../pymf_amd/csrc/pmf_dev.h"
#include "../pymf_amd/csrc/pmf_tiled.h"
#include "lab_chunk.h"
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
  // ---- round 5: the chunked forms (pmf_chunk.h) into buffers of their own; compared element by element below ----
  float *C2, *slab2;
  CK(hipMalloc(&C2, (size_t)m * KP * 4)); CK(hipMalloc(&slab2, (size_t)nch * KP * (n + KP) * 4));
  CK(hipMemset(C2, 0xff, (size_t)m * KP * 4)); CK(hipMemset(slab2, 0xff, (size_t)nch * KP * (n + KP) * 4));
  const bool chunk_ok = n % 1024 == 0 && (m / 16) % (4 * ROWGEMM_CHUNK_GB) == 0 && rpc % 16 == 0;
  CK(hipFuncSetAttribute((const void*)&k_colgemm_chunk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)colgemm_chunk_smem_bytes()));
  CK(hipFuncSetAttribute((const void*)&k_rowgemm_chunk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rowgemm_chunk_smem_bytes()));
  const int rg_groups = (int)(m / 16 / (4 * ROWGEMM_CHUNK_GB));
  auto run_row2 = [&] { hipLaunchKernelGGL(k_rowgemm_chunk, dim3((unsigned)std::min(rg_groups, 256)), dim3(256), rowgemm_chunk_smem_bytes(), 0, V, (int64_t)n, n, H, (int64_t)n, C2,
                                           (int64_t)KP, rg_groups); };
#ifdef PMF_CHUNK_STAMPS
  unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)nch * 4 * 2 * 8));
  auto run_col2 = [&] { hipLaunchKernelGGL(k_colgemm_chunk, dim3((unsigned)nch, (unsigned)(n / 1024)), dim3(256), colgemm_chunk_smem_bytes(), 0, V, (int64_t)n, n, W, (int64_t)KP, m, rpc, slab2,
                                           (int64_t)n + KP, dbg); };
#else
  auto run_col2 = [&] { hipLaunchKernelGGL(k_colgemm_chunk, dim3((unsigned)nch, (unsigned)(n / 1024)), dim3(256), colgemm_chunk_smem_bytes(), 0, V, (int64_t)n, n, W, (int64_t)KP, m, rpc, slab2,
                                           (int64_t)n + KP); };
#endif
  for (int which = 0; which < (chunk_ok ? 4 : 2); ++which) {
    auto run = [&] { if (which == 0) run_row(); else if (which == 1) run_col(); else if (which == 2) run_row2(); else run_col2(); };
    for (int w = 0; w < 20; ++w) run();
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < 20; ++r) { CK(hipEventRecord(e0)); run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms; }
    {   // the same kernel in a HOT loop (no host synchronisation between the launches: the clock the iteration's kernels run at)
      for (int w = 0; w < 200; ++w) run();
      CK(hipEventRecord(e0)); for (int w = 0; w < 100; ++w) run(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("   hot loop: %.4f ms per launch\n", ms / 100);
#ifdef PMF_CHUNK_STAMPS
      if (which == 3) {
        std::vector<unsigned long long> hd((size_t)nch * 8); CK(hipMemcpy(hd.data(), dbg, hd.size() * 8, hipMemcpyDeviceToHost));
        double cy = 0, wl = 0; for (int q = 0; q < nch * 4; ++q) { cy += hd[2 * q]; wl += hd[2 * q + 1]; }
        printf("   block loop of k_colgemm_chunk: %.0f cycles per block and wave, %.1f us per wave, %.3f GHz\n", cy / (nch * 4) / (rpc / 16), wl / (nch * 4) * 0.01, cy / (wl * 10.0));
      }
#endif
    }
    const double fl = (which & 1) == 0 ? 2.0 * m * n * KP : 2.0 * m * n * KP + 2.0 * m * KP * KP;
    const char* names[4] = {"k_rowgemm_stream<4,4,store>", "k_colgemm_stream<4,true>   ", "k_rowgemm_chunk            ", "k_colgemm_chunk            "};
    printf("%s  mean %.4f ms  best %.4f ms  %.1f TFLOP/s (mean)\n", names[which], sum / 20, best, fl / (sum / 20 * 1e-3) / 1e12);
  }
  if (chunk_ok) {
    std::vector<float> a((size_t)m * KP), b((size_t)m * KP);
    CK(hipMemcpy(a.data(), C, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), C2, b.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; double worst = 0; for (size_t q = 0; q < a.size(); ++q) { if (memcmp(&a[q], &b[q], 4)) { ++bad; worst = std::max(worst, (double)fabsf(a[q] - b[q]) / (fabsf(a[q]) + 1e-30)); } }
    printf("C: %zu of %zu elements differ between the stream and the chunked kernel (worst rel %.3g)\n", bad, a.size(), worst);
    std::vector<float> sa((size_t)nch * KP * (n + KP)), sb(sa.size());
    CK(hipMemcpy(sa.data(), slab, sa.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), slab2, sb.size() * 4, hipMemcpyDeviceToHost));
    bad = 0; worst = 0; for (size_t q = 0; q < sa.size(); ++q) { if (memcmp(&sa[q], &sb[q], 4)) { ++bad; worst = std::max(worst, (double)fabsf(sa[q] - sb[q]) / (fabsf(sa[q]) + 1e-30)); } }
    printf("slabs: %zu of %zu elements differ (worst rel %.3g)\n", bad, sa.size(), worst);
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
