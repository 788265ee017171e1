// gridsync_probe.hip -- what the hand-offs of the persistent one-pass kernel cost without its MFMA work:
// per "iteration" every workgroup stores a slab (sc1), crosses a grid barrier, sums its share of all
// slabs (pmf_reduce_slabs_dist), crosses a second barrier and loads the whole reduced slab into LDS.
// Every word is checked in every iteration (the slabs hold small integers: float sums are exact).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gridsync_probe.hip -o tools/gridsync_probe
//   tools/gridsync_probe [ntiles=74] [iters=200] [wgs=0 -> one per CU] [work_us=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pymf_amd/csrc/pmf_gridsync.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float slab_val(int wg, int it, int idx) { return (float)((wg & 15) + ((it + idx * 5 + (idx >> 6)) & 7)); }

__global__ __launch_bounds__(256, 1) void k_probe(float* slab, float* pst, unsigned* sync, int ntiles, int iters,
                                                  int work_ticks, unsigned long long* stamps, unsigned* errors) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int ok_lds;
  double* red = reinterpret_cast<double*>(smem);
  float* img = smem + 4096;                                // reduced slab image (ntiles KiB)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nwg = gridDim.x;
  unsigned bar = 0;
  float sumw = 0.f;
  for (int w = 0; w < nwg; ++w) sumw += (float)(w & 15);
  if (!pmf_grid_barrier(sync, ++bar, &ok_lds)) return;     // census
  const __amdgpu_buffer_rsrc_t rslab = pmf_rsrc(slab), rpst = pmf_rsrc(pst);
  unsigned long long acc[5] = {0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    unsigned long long t0 = pmf_realtime();
    if (work_ticks > 0) { while (pmf_realtime() - t0 < (unsigned long long)work_ticks) __builtin_amdgcn_s_sleep(8); }
    unsigned long long t1 = pmf_realtime();
    // slab out: wave w stores tiles w, w + 4, ...
    for (int t = wv; t < ntiles; t += 4) {
      const int idx = (t * 64 + lane) * 4;
      f32x4 v = {slab_val(blockIdx.x, it, idx), slab_val(blockIdx.x, it, idx + 1), slab_val(blockIdx.x, it, idx + 2),
                 slab_val(blockIdx.x, it, idx + 3)};
      pmf_st16_sc1(rslab, (unsigned)blockIdx.x * (unsigned)ntiles * 1024u + (unsigned)idx * 4u, v);
    }
    unsigned long long t2 = pmf_realtime();
    if (!pmf_grid_barrier(sync, ++bar, &ok_lds)) break;
    unsigned long long t3 = pmf_realtime();
    pmf_reduce_slabs_dist(slab, nwg, ntiles, pst, red, [](int, int, int, float) {});
    unsigned long long t4 = pmf_realtime();
    if (!pmf_grid_barrier(sync, ++bar, &ok_lds)) break;
    unsigned long long t5 = pmf_realtime();
    for (int q = tid; q < ntiles * 64; q += 256)
      *reinterpret_cast<f32x4*>(img + 4 * q) = pmf_ld16_sc1(rpst, (unsigned)q * 16u);
    __syncthreads();
    unsigned bad = 0;
    for (int q = tid; q < ntiles * 256; q += 256) {
      const float want = sumw + (float)nwg * (float)((it + q * 5 + (q >> 6)) & 7);
      if (img[q] != want) ++bad;
    }
    if (bad) atomicAdd(errors, bad);
    __syncthreads();
    unsigned long long t6 = pmf_realtime();
    acc[0] += t2 - t1; acc[1] += t3 - t2; acc[2] += t4 - t3; acc[3] += t5 - t4; acc[4] += t6 - t5;
  }
  if (tid == 0)
    for (int q = 0; q < 5; ++q) stamps[blockIdx.x * 5 + q] = acc[q];
  pmf_grid_leave(sync);
}

int main(int argc, char** argv) {
  const int ntiles = argc > 1 ? atoi(argv[1]) : 74;
  const int iters = argc > 2 ? atoi(argv[2]) : 200;
  int wgs = argc > 3 ? atoi(argv[3]) : 0;
  const int work_us = argc > 4 ? atoi(argv[4]) : 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  if (wgs <= 0) wgs = prop.multiProcessorCount;
  float *slab, *pst;
  unsigned *sync, *errors;
  unsigned long long* stamps;
  CHECK(hipMalloc(&slab, (size_t)wgs * ntiles * 1024));
  CHECK(hipMalloc(&pst, (size_t)ntiles * 1024));
  CHECK(hipMalloc(&sync, PMF_SYNC_WORDS * 4));
  CHECK(hipMalloc(&errors, 4));
  CHECK(hipMalloc(&stamps, (size_t)wgs * 5 * 8));
  CHECK(hipMemset(sync, 0, PMF_SYNC_WORDS * 4));
  CHECK(hipMemset(errors, 0, 4));
  const size_t smem = 160 * 1024 - 64;                     // one workgroup per CU, like the one-pass kernel
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_probe, dim3(wgs), dim3(256), smem, 0, slab, pst, sync, ntiles, iters, work_us * 100, stamps, errors);
    CHECK(hipGetLastError());
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)wgs * 5);
    unsigned herr = 0, hsync[PMF_SYNC_WORDS];
    CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hsync, sync, sizeof(hsync), hipMemcpyDeviceToHost));
    double mean[5] = {0, 0, 0, 0, 0}, mx[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < wgs; ++w)
      for (int q = 0; q < 5; ++q) {
        const double us = (double)h[(size_t)w * 5 + q] / 100.0 / iters;
        mean[q] += us / wgs;
        if (us > mx[q]) mx[q] = us;
      }
    printf("rep %d: wgs %d ntiles %d iters %d work %d us: %.2f us/iter (minus work %.2f) | errors %u abort %u counters-left %u\n", rep,
           wgs, ntiles, iters, work_us, ms * 1000.0 / iters, ms * 1000.0 / iters - work_us, herr, hsync[PMF_SYNC_ABORT], hsync[0] + hsync[PMF_SYNC_EXIT]);
    printf("   mean/max us per iter: slab store %.2f/%.2f  barrier1 %.2f/%.2f  reduce %.2f/%.2f  barrier2 %.2f/%.2f  load+check %.2f/%.2f\n",
           mean[0], mx[0], mean[1], mx[1], mean[2], mx[2], mean[3], mx[3], mean[4], mx[4]);
  }
  return 0;
}
