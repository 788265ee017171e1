// mall_lab.hip -- round 6, VERDICT r5 next 4: would cfg3's SECOND reader of a row chunk of V (k_colgemm_stream behind
// k_rowgemm_stream on the same rows) gain from finding the chunk in the 256 MiB Infinity Cache?
//
// The production stream kernels (pmf_tiled.h) at n = 1 024, k = 64 on sub-matrices of R rows (R x 4 KiB = 64 ... 192 MiB), every
// launch with the SAME grid and rows per workgroup, in two cache states:
//   resident : a hot loop over ONE sub-matrix -- its lines survive from launch to launch (chunk + outputs < 256 MiB)
//   rotating : launch i reads sub-matrix i mod P of a 2 GiB pool -- every line was last touched > 256 MiB of traffic ago (HBM)
// plus the pair the restructured W half step would run: rowgemm(c) then colgemm(c), chunk after chunk over the whole
// 262 144-row matrix, the colgemm launches timed apart, against the same launches with ALL rowgemms first (no reuse).
// And the plain read rate of the same sub-matrices in the two states (LDS-DMA, whole 4-KiB rows per wave).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mall_lab.hip -o tools/mall_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../pymf_amd/csrc/pmf_dev.h"
#include "../pymf_amd/csrc/pmf_tiled.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define GLDS16(gsrc, ldst)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc), \
                                   (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)
__global__ void k_fill(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (float)(x >> 8) * (1.0f / 16777216.0f);
  }
}
// whole 4-KiB rows per wave through LDS-DMA, 32 KiB in flight per wave (tools/read_bw_lab.hip pattern 3)
__global__ __launch_bounds__(256, 1) void k_read_rows(const float* __restrict__ V, int64_t m, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  float* sv = smem + wv * 8192;
  const char* Vb = reinterpret_cast<const char*>(V);
  const int gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
  auto issue = [&](int64_t it, int q, int buf) {
    const int64_t row = ((int64_t)gw + it * nw) * 4 + (q >> 2);
    GLDS16(Vb + (size_t)row * 4096 + (q & 3) * 1024 + 16 * lane, sv + buf * 4096 + q * 256);
  };
  const int64_t nit = m / 4 / nw;
#pragma unroll
  for (int q = 0; q < 16; ++q) issue(0, q, 0);
  for (int64_t it = 1; it < nit; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) issue(it, q, (int)(it & 1));
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && gw == 0) out[0] = sv[0];
}

int main() {
  constexpr int NT = 4, KP = 64, n = 1024;
  const int64_t M = 524288;                                  // the pool: 2 GiB of V
  float *V, *H, *W, *C, *slab, *out;
  CK(hipMalloc(&V, (size_t)M * n * 4)); CK(hipMalloc(&H, (size_t)KP * n * 4)); CK(hipMalloc(&W, (size_t)M * KP * 4)); CK(hipMalloc(&C, (size_t)M * KP * 4));
  CK(hipMalloc(&out, 64));
  k_fill<<<4096, 256>>>(V, (size_t)M * n, 1u); k_fill<<<256, 256>>>(H, (size_t)KP * n, 2u); k_fill<<<1024, 256>>>(W, (size_t)M * KP, 3u);
  const size_t slab_max = (size_t)1024 * KP * (n + KP);
  CK(hipMalloc(&slab, slab_max * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)&k_read_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 16384));
  const size_t ssm = rowgemm_stream_smem_bytes<NT, EPI_STORE, false>();
  const size_t csm = (size_t)2 * 64 * (16 * NT + 4) * sizeof(float);
  // one launch of each kind on rows [r0, r0 + R); rpc: rows per colgemm workgroup; RB: row blocks per rowgemm wave
  auto row4 = [&](int64_t r0, int64_t R) {
    const int ntiles = (int)(R / 64), ngroups = (ntiles + 3) / 4;
    hipLaunchKernelGGL((k_rowgemm_stream<NT, 4, EPI_STORE>), dim3((unsigned)std::min(ngroups, 512)), dim3(256), ssm, 0, V + (size_t)r0 * n, (int64_t)n, n, H, (int64_t)n,
                       (float*)nullptr, (const float*)nullptr, C + (size_t)r0 * KP, (int64_t)KP, 0.f, R, KP, ntiles, (int64_t)KP);
  };
  auto col = [&](int64_t r0, int64_t R, int rpc, float* sl) {
    const int nch = (int)((R + rpc - 1) / rpc);
    hipLaunchKernelGGL((k_colgemm_stream<NT, true>), dim3((unsigned)nch, 4u), dim3(256), csm, 0, V + (size_t)r0 * n, (int64_t)n, n, W + (size_t)r0 * KP, (int64_t)KP, R, rpc, sl,
                       (int64_t)n + KP, 0);
  };
  auto rd = [&](int64_t r0, int64_t R) { k_read_rows<<<256, 256, 4 * 2 * 16384>>>(V + (size_t)r0 * n, R, out); };
  printf("# part 1: one launch geometry, two cache states (mean of 40 launches in a hot loop; us per launch | TB/s of V)\n");
  printf("# %8s %8s | %-34s | %-34s | %-34s\n", "rows", "MiB", "plain read (resident | rotating)", "k_colgemm_stream (res | rot)", "k_rowgemm_stream<4,4> (res | rot)");
  for (int64_t R : {16384, 24576, 32768, 49152, 65536, 131072, 262144}) {
    const int P = (int)(M / R);                               // sub-matrices in the pool
    const int rpc = (int)std::max<int64_t>(64, ((R / 16 + 255) / 256 + 3) / 4 * 4 * 16);   // the library's rule: ~256 row chunks
    double t[3][2];
    for (int kind = 0; kind < 3; ++kind)
      for (int rot = 0; rot < 2; ++rot) {
        auto launch = [&](int i) { const int64_t r0 = rot ? (int64_t)(i % P) * R : 0; if (kind == 0) rd(r0, R); else if (kind == 1) col(r0, R, rpc, slab); else row4(r0, R); };
        for (int i = 0; i < 40; ++i) launch(i);
        CK(hipEventRecord(e0)); for (int i = 0; i < 40; ++i) launch(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t[kind][rot] = ms / 40 * 1e3;
      }
    const double by = (double)R * n * 4;
    printf("  %8lld %8.0f | %7.1f %5.2f | %7.1f %5.2f       | %7.1f %5.2f | %7.1f %5.2f       | %7.1f %5.2f | %7.1f %5.2f\n", (long long)R, by / 1048576.0,
           t[0][0], by / t[0][0] * 1e-6, t[0][1], by / t[0][1] * 1e-6, t[1][0], by / t[1][0] * 1e-6, t[1][1], by / t[1][1] * 1e-6, t[2][0], by / t[2][0] * 1e-6, t[2][1], by / t[2][1] * 1e-6);
  }
  printf("# part 2: the whole 262 144-row matrix as chunks -- rowgemm(c) then colgemm(c) [pair] against all rowgemms, then all colgemms [apart];\n");
  printf("#         sum of the launches' own event pairs, us for the whole matrix (unchunked production launches: last line)\n");
  const int64_t m = 262144;
  std::vector<hipEvent_t> ev(4 * 64);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int64_t R : {16384, 24576, 32768, 65536, 262144}) {
    const int nc = (int)((m + R - 1) / R);
    const int rpc = (int)std::max<int64_t>(64, ((R / 16 + 255) / 256 + 3) / 4 * 4 * 16);
    const size_t per = (size_t)((R + rpc - 1) / rpc) * KP * (n + KP);
    for (int mode = 0; mode < 2; ++mode) {                   // 0: pair, 1: apart
      double tr = 0, tc = 0, wall = 0;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) {
          for (int c = 0; c < nc; ++c) {
            const int64_t r0 = (int64_t)c * R, Rc = std::min(R, m - r0);
            CK(hipEventRecord(ev[4 * c])); row4(r0, Rc); CK(hipEventRecord(ev[4 * c + 1]));
            col(r0, Rc, rpc, slab + (per * c) % (slab_max - per + 1)); CK(hipEventRecord(ev[4 * c + 2]));
          }
        } else {
          for (int c = 0; c < nc; ++c) { const int64_t r0 = (int64_t)c * R, Rc = std::min(R, m - r0); CK(hipEventRecord(ev[4 * c])); row4(r0, Rc); CK(hipEventRecord(ev[4 * c + 1])); }
          for (int c = 0; c < nc; ++c) { const int64_t r0 = (int64_t)c * R, Rc = std::min(R, m - r0); CK(hipEventRecord(ev[4 * c + 3])); col(r0, Rc, rpc, slab + (per * c) % (slab_max - per + 1)); CK(hipEventRecord(ev[4 * c + 2])); }
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        if (rep < 2) continue;
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); wall += ms * 1e3 / 4;
        for (int c = 0; c < nc; ++c) {
          CK(hipEventElapsedTime(&ms, ev[4 * c], ev[4 * c + 1])); tr += ms * 1e3 / 4;
          CK(hipEventElapsedTime(&ms, mode == 0 ? ev[4 * c + 1] : ev[4 * c + 3], ev[4 * c + 2])); tc += ms * 1e3 / 4;
        }
      }
      printf("  chunk %7lld rows (%4.0f MiB) x %2d  %-5s  rowgemm %7.1f  colgemm %7.1f  wall %7.1f\n", (long long)R, (double)R * n * 4 / 1048576.0, nc, mode ? "apart" : "pair", tr, tc, wall);
    }
  }
  return 0;
}
