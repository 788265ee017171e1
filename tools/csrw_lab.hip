// csrw_lab.hip -- W = V M for CSR V at cfg5's shape (4 194 304 x 128 at 1 % nnz, k = 128): how close to the plain-store
// ceiling (6.0-6.2 TB/s, MI355X_MICROARCH.md) can the W write get?  Variants of k_csr_w_blocks<8> (pmf_csr.h) side by side.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/csrw_lab.hip -o tools/csrw_lab && tools/csrw_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../pymf_amd/csrc/pmf_csr.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// pure store stream of the same shape: what the memory system takes with no CSR work at all
template <int MODE>   // 0 nt, 1 plain
__global__ __launch_bounds__(1024) void k_store_only(float* __restrict__ W, int64_t nblk) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t nwaves = (int64_t)gridDim.x * 16;
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int64_t blk = (int64_t)blockIdx.x * 16 + wv; blk < nblk; blk += nwaves) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      f32x4* p = reinterpret_cast<f32x4*>(W + (size_t)blk * 16 * 128 + (size_t)g * 256 + 4 * lane);
      if (MODE == 0) __builtin_nontemporal_store(v, p); else *p = v;
    }
  }
}

// linear (memset-like) order: at step s thread t of the whole grid stores 16 bytes at (s * threads + t) * 16
template <int MODE, int TPB>
__global__ __launch_bounds__(TPB) void k_store_linear(float* __restrict__ W, int64_t n16) {
  const int64_t nthreads = (int64_t)gridDim.x * TPB;
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (int64_t q = (int64_t)blockIdx.x * TPB + threadIdx.x; q < n16; q += nthreads) {
    f32x4* p = reinterpret_cast<f32x4*>(W) + q;
    if (MODE == 0) __builtin_nontemporal_store(v, p); else *p = v;
  }
}
// block order of k_csr_w_blocks with TPB threads per workgroup
template <int MODE, int TPB>
__global__ __launch_bounds__(TPB) void k_store_blocks(float* __restrict__ W, int64_t nblk) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t nwaves = (int64_t)gridDim.x * (TPB / 64);
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int64_t blk = (int64_t)blockIdx.x * (TPB / 64) + wv; blk < nblk; blk += nwaves) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      f32x4* p = reinterpret_cast<f32x4*>(W + (size_t)blk * 16 * 128 + (size_t)g * 256 + 4 * lane);
      if (MODE == 0) __builtin_nontemporal_store(v, p); else *p = v;
    }
  }
}

// software-pipelined variant: the row pointers and the (column, value) run of the wave's NEXT block are requested before the
// current block's eight stores are formed
template <int NT>
__global__ __launch_bounds__(1024) void k_csr_w_pipe(const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                     const float* __restrict__ vals, int64_t nblk, int np,
                                                     const float* __restrict__ M, float* __restrict__ W) {
  constexpr int KP = 16 * NT, LPR = KP / 4, RPI = 64 / LPR;
  extern __shared__ __attribute__((aligned(16))) float sMw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int q = tid; q < np * KP / 4; q += 1024) reinterpret_cast<f32x4*>(sMw)[q] = reinterpret_cast<const f32x4*>(M)[q];
  __syncthreads();
  const int sub = lane / LPR, c4 = 4 * (lane % LPR);
  const int64_t nwaves = (int64_t)gridDim.x * 16;
  int64_t blk = (int64_t)blockIdx.x * 16 + wv;
  if (blk >= nblk) return;
  long long ipv = (long long)indptr[blk * 16 + (lane < 17 ? lane : 16)];
  long long a = __shfl(ipv, 0, 64);
  int rel = (int)(ipv - a);
  int nzb = __shfl(rel, 16, 64);
  int colv = 0; float valv = 0.f;
  if (lane < nzb) { colv = indices[a + lane]; valv = vals[a + lane]; }
  for (; blk < nblk; blk += nwaves) {
    const int64_t r0 = blk * 16, nb = blk + nwaves;
    // next block's pointers first (two dependent requests in flight under this block's work)
    long long ipv_n = 0, a_n = 0; int rel_n = 0, nzb_n = 0, colv_n = 0; float valv_n = 0.f;
    if (nb < nblk) ipv_n = (long long)indptr[nb * 16 + (lane < 17 ? lane : 16)];
#pragma unroll
    for (int g = 0; g < 16 / RPI; ++g) {
      const int row = g * RPI + sub;
      const int ea = __shfl(rel, row, 64), eb = __shfl(rel, row + 1, 64);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int cnt = eb - ea;
      int cmax = cnt;
#pragma unroll
      for (int o = 32; o >= LPR; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o, 64));
      for (int it = 0; it < cmax; ++it) {
        const int e = ea + it;
        int col = __shfl(colv, e & 63, 64);
        float val = __shfl(valv, e & 63, 64);
        if (it < cnt && e >= 64) { col = indices[a + e]; val = vals[a + e]; }
        if (it >= cnt) { col = 0; val = 0.f; }
        const f32x4 mrow = *reinterpret_cast<const f32x4*>(sMw + (size_t)col * KP + c4);
        acc += val * mrow;
      }
      __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(W + (size_t)(r0 + row) * KP + c4));
      if (g == 1 && nb < nblk) {                       // the pointers have landed by now: request the entries
        a_n = __shfl(ipv_n, 0, 64);
        rel_n = (int)(ipv_n - a_n);
        nzb_n = __shfl(rel_n, 16, 64);
        if (lane < nzb_n) { colv_n = indices[a_n + lane]; valv_n = vals[a_n + lane]; }
      }
    }
    a = a_n; rel = rel_n; nzb = nzb_n; colv = colv_n; valv = valv_n;
  }
}

int main() {
  const int64_t m = 4194304; const int n = 128, KP = 128;
  std::mt19937_64 rng(1234);
  std::poisson_distribution<int> pois(1.28);
  std::vector<int64_t> ip(m + 1, 0);
  for (int64_t r = 0; r < m; ++r) ip[r + 1] = ip[r] + std::min(pois(rng), n);
  const int64_t nnz = ip[m];
  std::vector<int32_t> ix(nnz); std::vector<float> vv(nnz), M((size_t)n * KP);
  for (int64_t e = 0; e < nnz; ++e) { ix[e] = (int32_t)(rng() % n); vv[e] = (float)(rng() % 1000) * 1e-3f; }
  for (auto& x : M) x = (float)(rng() % 2000) * 1e-3f - 1.f;
  int64_t* dip; int32_t* dix; float *dvv, *dM, *dW, *dW2;
  CK(hipMalloc(&dip, (m + 1) * 8)); CK(hipMalloc(&dix, nnz * 4)); CK(hipMalloc(&dvv, nnz * 4)); CK(hipMalloc(&dM, M.size() * 4));
  CK(hipMalloc(&dW, (size_t)m * KP * 4)); CK(hipMalloc(&dW2, (size_t)m * KP * 4));
  CK(hipMemcpy(dip, ip.data(), (m + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dix, ix.data(), nnz * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dvv, vv.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dM, M.data(), M.size() * 4, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csr_w_blocks<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csr_w_pipe<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  const int64_t nblk = m / 16;
  const double bytes = 4.0 * m * KP + 8.0 * nnz + 8.0 * (m + 1);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch, double by) {
    for (int w = 0; w < 3; ++w) launch();
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0.f;
    for (int r = 0; r < 10; ++r) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms); sum += ms;
    }
    printf("%-44s mean %.4f ms  best %.4f ms  %.2f TB/s (mean)\n", name, sum / 10, best, by / (sum / 10 * 1e-3) / 1e12);
  };
  const size_t smem = (size_t)n * KP * 4;
  for (int wgs : {512, 480, 384, 256, 1024}) {
    char nm[96];
    snprintf(nm, sizeof(nm), "k_csr_w_blocks<8>, %d workgroups", wgs);
    timeit(nm, [&] { hipLaunchKernelGGL((k_csr_w_blocks<8>), dim3(wgs), dim3(1024), smem, 0, dip, dix, dvv, nblk, n, dM, dW, 1); }, bytes);
  }
  for (int wgs : {256, 512, 2048, 16384}) {
    char nm[96];
    snprintf(nm, sizeof(nm), "k_csr_w_blocks<8>, M from L2, %d workgroups", wgs);
    timeit(nm, [&] { hipLaunchKernelGGL((k_csr_w_blocks<8>), dim3(wgs), dim3(1024), 0, 0, dip, dix, dvv, nblk, n, dM, dW, 0); }, bytes);
  }
  for (int wgs : {512, 480, 384}) {
    char nm[96];
    snprintf(nm, sizeof(nm), "pipelined loads, %d workgroups", wgs);
    timeit(nm, [&] { hipLaunchKernelGGL((k_csr_w_pipe<8>), dim3(wgs), dim3(1024), smem, 0, dip, dix, dvv, nblk, n, dM, dW2); }, bytes);
  }
  for (int wgs : {512, 256, 1024, 2048}) {
    char nm[96];
    snprintf(nm, sizeof(nm), "stores only (nt), %d workgroups", wgs);
    timeit(nm, [&] { hipLaunchKernelGGL((k_store_only<0>), dim3(wgs), dim3(1024), 0, 0, dW2, nblk); }, 4.0 * m * KP);
    snprintf(nm, sizeof(nm), "stores only (plain), %d workgroups", wgs);
    timeit(nm, [&] { hipLaunchKernelGGL((k_store_only<1>), dim3(wgs), dim3(1024), 0, 0, dW2, nblk); }, 4.0 * m * KP);
  }
  {
    const int64_t n16 = (int64_t)m * KP / 4;
    char nm[96];
    for (int wgs : {256, 512, 1024, 2048, 8192}) {
      snprintf(nm, sizeof(nm), "linear order, 1024 thr x %d (nt)", wgs);
      timeit(nm, [&] { hipLaunchKernelGGL((k_store_linear<0, 1024>), dim3(wgs), dim3(1024), 0, 0, dW2, n16); }, 4.0 * m * KP);
      snprintf(nm, sizeof(nm), "linear order, 1024 thr x %d (plain)", wgs);
      timeit(nm, [&] { hipLaunchKernelGGL((k_store_linear<1, 1024>), dim3(wgs), dim3(1024), 0, 0, dW2, n16); }, 4.0 * m * KP);
    }
    for (int wgs : {1024, 2048, 4096, 65536}) {
      snprintf(nm, sizeof(nm), "linear order, 256 thr x %d (nt)", wgs);
      timeit(nm, [&] { hipLaunchKernelGGL((k_store_linear<0, 256>), dim3(wgs), dim3(256), 0, 0, dW2, n16); }, 4.0 * m * KP);
      snprintf(nm, sizeof(nm), "linear order, 256 thr x %d (plain)", wgs);
      timeit(nm, [&] { hipLaunchKernelGGL((k_store_linear<1, 256>), dim3(wgs), dim3(256), 0, 0, dW2, n16); }, 4.0 * m * KP);
    }
    for (int wgs : {256, 512, 1024}) {
      snprintf(nm, sizeof(nm), "block order, 512 thr x %d (nt)", wgs);
      timeit(nm, [&] { hipLaunchKernelGGL((k_store_blocks<0, 512>), dim3(wgs), dim3(512), 0, 0, dW2, nblk); }, 4.0 * m * KP);
      snprintf(nm, sizeof(nm), "block order, 256 thr x %d (nt)", wgs);
      timeit(nm, [&] { hipLaunchKernelGGL((k_store_blocks<0, 256>), dim3(wgs), dim3(256), 0, 0, dW2, nblk); }, 4.0 * m * KP);
      snprintf(nm, sizeof(nm), "block order, 256 thr x %d (plain)", wgs);
      timeit(nm, [&] { hipLaunchKernelGGL((k_store_blocks<1, 256>), dim3(wgs), dim3(256), 0, 0, dW2, nblk); }, 4.0 * m * KP);
    }
  }
  timeit("hipMemsetAsync of W", [&] { CK(hipMemsetAsync(dW2, 0, (size_t)m * KP * 4, 0)); }, 4.0 * m * KP);
  // correctness of the pipelined variant
  hipLaunchKernelGGL((k_csr_w_blocks<8>), dim3(512), dim3(1024), smem, 0, dip, dix, dvv, nblk, n, dM, dW, 1);
  hipLaunchKernelGGL((k_csr_w_pipe<8>), dim3(512), dim3(1024), smem, 0, dip, dix, dvv, nblk, n, dM, dW2);
  CK(hipDeviceSynchronize());
  std::vector<float> a((size_t)1 << 22), b((size_t)1 << 22);
  size_t bad = 0;
  for (size_t off : {(size_t)0, (size_t)m * KP / 2, (size_t)m * KP - a.size()}) {
    CK(hipMemcpy(a.data(), dW + off, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dW2 + off, b.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < a.size(); ++i) bad += a[i] != b[i];
  }
  printf("pipelined variant vs k_csr_w_blocks: %zu differing entries in 3 x 4M samples\n", bad);
  return 0;
}
