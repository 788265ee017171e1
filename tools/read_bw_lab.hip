// read_bw_lab.hip -- how fast can one workgroup per CU (4 waves, one per SIMD) pull a row-major matrix through LDS-DMA, by access
// pattern?  (round 5: the two long products of cfg3 sit at 3.5-3.8 TB/s of V; is that the pattern or the chip?)
//   V [m][n] float32, n = 1024 (4 KiB rows) or 256 (1 KiB rows); requests of 64 lanes x 16 B = 1 KiB, `depth` blocks of 16 requests in flight.
//   pattern 0: a wave streams CONTIGUOUS 16-KiB blocks (16 rows x 1 KiB rows: the one-pass kernel at 256 columns)
//   pattern 1: 16 rows x 1 KiB pieces of 4-KiB rows, the 4 waves of a workgroup on the 4 column chunks of the SAME rows (k_colgemm_chunk)
//   pattern 2: as 1, but every wave on rows of its own (k_rowgemm_chunk's chunk-major sweep)
//   pattern 3: 4-KiB rows, a request = ONE row's 1 KiB ... a wave takes whole rows: 4 requests per row (row-contiguous 4 KiB)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define GLDS16(gsrc, ldst)                                                                \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc), \
                                   (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)

template <int PAT>
__global__ __launch_bounds__(256, 1) void k_read(const float* __restrict__ V, int64_t m, int n, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];     // 4 waves x 2 x 16 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* sv = smem + wv * 8192;
  const char* Vb = reinterpret_cast<const char*>(V);
  const int64_t nblk = m / 16;                                     // 16-row blocks
  const int gw = blockIdx.x * 4 + wv, nw = gridDim.x * 4;
  // per-lane byte offset inside a request, and the request's base, by pattern
  const size_t pitch = (size_t)n * 4;
  auto issue = [&](int64_t it, int q, int buf) {                   // request q (0..15) of this wave's it-th block
    const char* src;
    if (PAT == 0) {                                                // contiguous 16 KiB
      const int64_t blk = (int64_t)gw + it * nw;
      src = Vb + (size_t)blk * 16384 + q * 1024 + 16 * lane;
    } else if (PAT == 1 || PAT == 2) {                             // 4 rows x 256 B of a 1-KiB column chunk
      const int64_t blk = PAT == 1 ? (int64_t)blockIdx.x + it * gridDim.x : (int64_t)gw + (it >> 2) * nw;   // 16-row block
      const int chunk = PAT == 1 ? wv : (int)(it & 3);
      const int p = q >> 2, rg = q & 3, row = 4 * rg + (lane >> 4);
      src = Vb + ((size_t)blk * 16 + row) * pitch + chunk * 1024 + p * 256 + 16 * (lane & 15);
    } else {                                                       // whole rows: request q = quarter (q & 3) of row 4 * it' ...
      const int64_t row = ((int64_t)gw + it * nw) * 4 + (q >> 2);  // 4 rows per "block" of 16 requests
      src = Vb + (size_t)row * pitch + (q & 3) * 1024 + 16 * lane;
    }
    GLDS16(src, sv + buf * 4096 + q * 256);
  };
  const int64_t units = PAT == 0 ? nblk * (n / 256) / ((n / 256)) : 0;   // (unused)
  (void)units;
  int64_t nit;
  if (PAT == 0) nit = (m * (int64_t)n * 4 / 16384) / nw;
  else if (PAT == 1) nit = nblk / gridDim.x;
  else if (PAT == 2) nit = nblk * 4 / nw;
  else nit = m / 4 / nw;
#pragma unroll
  for (int q = 0; q < 16; ++q) issue(0, q, 0);
  for (int64_t it = 1; it < nit; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) issue(it, q, (int)(it & 1));
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");              // the block before has landed
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && gw == 0) out[0] = sv[0];
}

template <int PAT>
int run(const float* V, int64_t m, int n, float* out, const char* what) {
  const size_t smem = 4 * 2 * 16384;
  CK(hipFuncSetAttribute((const void*)&k_read<PAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int it = 0; it < 30; ++it) k_read<PAT><<<256, 256, smem>>>(V, m, n, out);
  CK(hipEventRecord(e0));
  for (int it = 0; it < 30; ++it) k_read<PAT><<<256, 256, smem>>>(V, m, n, out);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  %-92s %.4f ms = %.2f TB/s\n", what, ms / 30, (double)m * n * 4 / (ms / 30 * 1e-3) / 1e12);
  return 0;
}

int main() {
  float *V, *out;
  const size_t bytes = (size_t)1 << 30;
  CK(hipMalloc(&V, bytes)); CK(hipMemset(V, 0x3c, bytes)); CK(hipMalloc(&out, 64));
  printf("1 GiB through LDS-DMA, 256 workgroups x 4 waves, 16-32 KiB in flight per wave\n");
  if (run<0>(V, 1048576, 256, out, "contiguous 16-KiB blocks per wave (one-pass kernel, 256 columns)")) return 1;
  if (run<1>(V, 262144, 1024, out, "16 rows x 1 KiB of 4-KiB rows, the 4 waves on the 4 chunks of the same rows (k_colgemm_chunk)")) return 1;
  if (run<2>(V, 262144, 1024, out, "16 rows x 1 KiB of 4-KiB rows, every wave on rows of its own, chunk after chunk")) return 1;
  if (run<3>(V, 262144, 1024, out, "whole 4-KiB rows per wave (4 requests per row)")) return 1;
  return 0;
}
