// A/B harness of the production one-pass kernel: hipcc -I <dir with pmf_fused.h> -DAB_NT=2 -DAB_NPANEL=4 -DAB_SPLIT=2 -DAB_NP=512 ...
// prints time per launch and checksums of W and of the summed slabs (same summation order => same bits).
#define PMF_FUSED_KERNEL_ONLY
#include "pmf_fused.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#ifndef AB_MODE
#define AB_MODE 0
#endif
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void fillk(float* p, size_t n, unsigned seed){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; if(i<n) p[i]=u01_from(seed,i); }
int main(int argc, char** argv){
  const int64_t mp = argc > 1 ? atoll(argv[1]) : 65536; const int NP=AB_NP, KP=16*AB_NT;
  const int per = AB_SPLIT==2 ? 2 : 4;
  int wgs = (int)std::min<int64_t>(256, (mp/16+per-1)/per);
  float *V,*W0,*W,*H,*G,*slab;
  CK(hipMalloc(&V,mp*NP*4)); CK(hipMalloc(&W0,mp*KP*4)); CK(hipMalloc(&W,mp*KP*4)); CK(hipMalloc(&H,KP*NP*4)); CK(hipMalloc(&G,KP*KP*4));
  const size_t slab_floats=(size_t)KP*(NP+KP); CK(hipMalloc(&slab,(size_t)wgs*slab_floats*4)); CK(hipMemset(slab,0,(size_t)wgs*slab_floats*4));
  fillk<<<(mp*NP+255)/256,256>>>(V,mp*NP,1); fillk<<<(mp*KP+255)/256,256>>>(W0,mp*KP,2);
  fillk<<<(KP*NP+255)/256,256>>>(H,KP*NP,3); fillk<<<(KP*KP+255)/256,256>>>(G,KP*KP,4);
  const size_t smem=fused_smem_bytes<AB_NT,AB_NPANEL,AB_SPLIT>();
  CK(hipFuncSetAttribute((const void*)&k_nmf_fused<AB_NT,AB_NPANEL,AB_MODE,AB_SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize,(int)smem));
  const int nblk=(int)(mp/16), nw=wgs*per, blk_per=nblk/nw, blk_extra=nblk%nw;
  const FusedCtl ctl{nullptr,nullptr,nullptr,0.0,0.0,0.0,0,-1};
  auto run=[&](){ k_nmf_fused<AB_NT,AB_NPANEL,AB_MODE,AB_SPLIT><<<wgs,256,smem>>>(V,W,H,G,blk_per,blk_extra,0.05f,slab,ctl,0); };
  CK(hipMemcpy(W,W0,mp*KP*4,hipMemcpyDeviceToDevice)); run(); CK(hipDeviceSynchronize());
  {
    std::vector<float> a((size_t)mp*KP); CK(hipMemcpy(a.data(),W,a.size()*4,hipMemcpyDeviceToHost));
    unsigned long long hsh=1469598103934665603ull; for(float x: a){ unsigned u; memcpy(&u,&x,4); hsh=(hsh^u)*1099511628211ull; }
    const size_t ntp = AB_SPLIT*AB_NPANEL*4, ntu=(size_t)AB_NT*ntp+AB_NT*(AB_NT+1)/2, pe=ntu*256;
    std::vector<float> s1((size_t)wgs*pe); CK(hipMemcpy(s1.data(),slab,s1.size()*4,hipMemcpyDeviceToHost));
    unsigned long long h2=1469598103934665603ull; for(float x: s1){ unsigned u; memcpy(&u,&x,4); h2=(h2^u)*1099511628211ull; }
    printf("W fnv %016llx  slabs fnv %016llx\n", hsh, h2);
  }
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for(int rep=0; rep<3; ++rep){
    (void)hipEventRecord(e0);
    for(int it=0; it<100; ++it) run();
    (void)hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; (void)hipEventElapsedTime(&ms,e0,e1);
    printf("k_nmf_fused<%d,%d,%d,%d> %lld x %d: %.2f us per launch\n", AB_NT,AB_NPANEL,AB_MODE,AB_SPLIT,(long long)mp,NP, ms/100*1e3);
  }
  return 0;
}
