#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 64 * 4);
  for (int rep = 0; rep < 3; ++rep) {
    k<<<1, 1024>>>(d); unsigned h[16]; hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
    for (int w = 0; w < 16; ++w) printf("w%d: wave %u simd %u cu %u | ", w, h[w] & 15, (h[w] >> 4) & 3, (h[w] >> 8) & 15);
    printf("\n");
  }
  return 0;
}
