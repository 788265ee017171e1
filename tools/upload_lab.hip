// upload_lab.hip -- how fast can 1 GiB of pageable host memory reach HBM?  
//   hipcc --offload-arch=gfx950 -O2 tools/upload_lab.hip -o tools/upload_lab && tools/upload_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t rows = 1048576, cols = 256, bytes = rows * cols * 4;
  float* h = (float*)malloc(bytes);
  for (size_t i = 0; i < rows * cols; i += 1024) h[i] = (float)i;      // touch every page
  float* d; CK(hipMalloc(&d, bytes));
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  CK(hipMemsetAsync(d, 0, bytes, s)); CK(hipStreamSynchronize(s));
  for (int rep = 0; rep < 2; ++rep) {
    double t = now();
    CK(hipMemcpy2DAsync(d, cols * 4, h, cols * 4, cols * 4, rows, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    printf("one hipMemcpy2DAsync, pageable:        %7.1f ms  %.1f GB/s\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
    t = now();
    CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    printf("one hipMemcpyAsync, pageable:          %7.1f ms  %.1f GB/s\n", (now() - t) * 1e3, bytes / (now() - t) / 1e9);
    for (size_t chunk : {(size_t)16 << 20, (size_t)64 << 20}) {
      t = now();
      for (size_t o = 0; o < bytes; o += chunk) CK(hipMemcpyAsync((char*)d + o, (char*)h + o, chunk, hipMemcpyHostToDevice, s));
      CK(hipStreamSynchronize(s));
      printf("hipMemcpyAsync in %3zu MiB chunks:       %7.1f ms  %.1f GB/s\n", chunk >> 20, (now() - t) * 1e3, bytes / (now() - t) / 1e9);
    }
    // pinned staging ring: host threads memcpy into pinned buffers, DMA from there
    for (int nthreads : {1, 4, 8}) {
      const size_t chunk = (size_t)32 << 20; const int NB = 4;
      static void* pin[4] = {nullptr, nullptr, nullptr, nullptr}; static hipEvent_t ev[4];
      if (!pin[0]) for (int b = 0; b < NB; ++b) { CK(hipHostMalloc(&pin[b], chunk, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming)); }
      t = now();
      int i = 0;
      for (size_t o = 0; o < bytes; o += chunk, ++i) {
        const int b = i % NB;
        if (i >= NB) CK(hipEventSynchronize(ev[b]));
        std::vector<std::thread> th;
        const size_t per = chunk / nthreads;
        for (int q = 0; q < nthreads; ++q) th.emplace_back([=] { memcpy((char*)pin[b] + q * per, (char*)h + o + q * per, per); });
        for (auto& x : th) x.join();
        CK(hipMemcpyAsync((char*)d + o, pin[b], chunk, hipMemcpyHostToDevice, s));
        CK(hipEventRecord(ev[b], s));
      }
      CK(hipStreamSynchronize(s));
      printf("pinned ring 4 x 32 MiB, %d copy threads: %7.1f ms  %.1f GB/s\n", nthreads, (now() - t) * 1e3, bytes / (now() - t) / 1e9);
    }
    t = now();
    CK(hipHostRegister(h, bytes, hipHostRegisterDefault));
    const double treg = now() - t;
    CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s));
    const double tall = now() - t;
    CK(hipHostUnregister(h));
    printf("hipHostRegister %.1f ms + copy %.1f ms = %7.1f ms  %.1f GB/s (unregister %.1f ms)\n", treg * 1e3, (tall - treg) * 1e3, tall * 1e3, bytes / tall / 1e9,
           (now() - t - tall) * 1e3);
  }
  // float64 -> float32 on the device vs on the host (W, H are float64 by default)
  const size_t wn = rows * 64;
  double* hw = (double*)malloc(wn * 8);
  for (size_t i = 0; i < wn; ++i) hw[i] = (double)i * 1e-9;
  double t = now();
  float* hf = (float*)malloc(wn * 4);
  for (size_t i = 0; i < wn; ++i) hf[i] = (float)hw[i];
  printf("host float64 -> float32 of %zu MiB (1 thread): %.1f ms\n", (wn * 8) >> 20, (now() - t) * 1e3);
  double* dw; CK(hipMalloc(&dw, wn * 8));
  t = now();
  for (size_t o = 0; o < wn * 8; o += (size_t)64 << 20) CK(hipMemcpyAsync((char*)dw + o, (char*)hw + o, std::min<size_t>((size_t)64 << 20, wn * 8 - o), hipMemcpyHostToDevice, s));
  CK(hipStreamSynchronize(s));
  printf("upload of the float64 W itself (512 MiB, 64 MiB chunks): %.1f ms\n", (now() - t) * 1e3);
  return 0;
}
