// diagnostic: time and accuracy of k_inverse_spd_mfma (pmf_inv.h) on a random Gram matrix H H^T, H k x 128 uniform
// build: hipcc --offload-arch=gfx950 -O3 -I pymf_amd/csrc tools/stamp_inv.hip -o /tmp/stamp_inv ; run: /tmp/stamp_inv [k]
#include <hip/hip_runtime.h>

#include <vector>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#ifndef PMF_INV_HEADER
#define PMF_INV_HEADER "/root/repo/pymf_amd/csrc/pmf_inv.h"
#endif
#include PMF_INV_HEADER
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
template <int NBLK>
int run(int k) {
  const int KP = 16 * NBLK, n = 128;
  std::vector<double> H((size_t)k*n), G((size_t)KP*KP, 0.0);
  srand(1); for (auto& x : H) x = rand() / (double)RAND_MAX;
  for (int i=0;i<KP;++i) for (int j=0;j<KP;++j){ double s=0; if(i<k&&j<k){ for(int t=0;t<n;++t) s+=H[(size_t)i*n+t]*H[(size_t)j*n+t]; if(i==j) s+=1e-3; } else s = (i==j); G[(size_t)i*KP+j]=s; }
  double *dG,*dI,*dP; int* dFlag; CK(hipMalloc(&dP,KP*KP*8)); CK(hipMalloc(&dFlag,4));
  CK(hipMalloc(&dG,KP*KP*8)); CK(hipMalloc(&dI,KP*KP*8));
  CK(hipMemcpy(dG,G.data(),KP*KP*8,hipMemcpyHostToDevice));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep=0; rep<3; ++rep){
    hipEventRecord(e0);
    for (int it=0; it<200; ++it) hipLaunchKernelGGL((k_inverse_spd_mfma<NBLK>), dim3(1), dim3(64 * NBLK * (NBLK / 4)), 0, 0, dG, KP, k, dI, (const int*)nullptr, (int*)nullptr, dFlag, getenv("INV_PATCH") ? dP : (double*)nullptr);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms,e0,e1);
    printf("k_inverse_spd_mfma<%d>, k = %d: %.2f us per launch (back to back)\n", NBLK, k, ms/200*1e3);
  }
  std::vector<double> I((size_t)KP*KP); CK(hipMemcpy(I.data(),dI,KP*KP*8,hipMemcpyDeviceToHost));
  double worst=0; for (int i=0;i<KP;++i) for(int j=0;j<KP;++j){ double s=0; for(int t=0;t<KP;++t) s+=G[(size_t)i*KP+t]*I[(size_t)t*KP+j]; const double e=fabs(s-(i==j)); if(!(e<=worst)) worst=e; }
#ifdef PMF_INV_STAMPS
  {
    const int nw = NBLK * (NBLK / 4), ns = (k + 15) / 16;
    std::vector<unsigned long long> d(16 * 8 * 4);
    CK(hipMemcpyFromSymbol(d.data(), HIP_SYMBOL(g_inv_dbg), d.size() * 8));
    const unsigned long long t0 = d[0];
    printf("stamps (shader cycles since wave 0 passed the first barrier): per step, [wave: after barrier 1 | R formed | after barrier 2 | step done]\n");
    for (int p = 0; p < ns; ++p) {
      printf(" step %d:", p);
      for (int w = 0; w < nw; ++w) { const unsigned long long* q = &d[(w * 8 + p) * 4]; printf("  w%d %llu|%llu|%llu|%llu", w, q[0] - t0, q[1] - t0, q[2] - t0, q[3] - t0); }
      printf("\n");
    }
  }
#endif
  unsigned long long h = 1469598103934665603ull;
  for (double x : I) { unsigned long long b; memcpy(&b, &x, 8); h = (h ^ b) * 1099511628211ull; }
  printf("max |G inv(G) - I| = %.3e   bits of the inverse: %016llx\n", worst, h);
  return 0;
}
int main(int argc, char** argv){
  const int k = argc > 1 ? atoi(argv[1]) : 128;
  return k <= 64 ? run<4>(k) : run<8>(k);
}
