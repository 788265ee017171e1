// diagnostic: per-section cycle shares of k_nmf_fused8<4, 0> (1,048,576 x 256, k = 128)
#define PMF_STAMPS
#define PMF_FUSED_KERNEL_ONLY
#include <algorithm>
#include "/root/repo/pymf_amd/csrc/pmf_coop.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void fillk(float* p, size_t n, unsigned seed){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; if(i<n) p[i]=u01_from(seed,i); }
int main(int argc, char** argv){
  const int64_t mp = argc > 1 ? atoll(argv[1]) : 1048576; const int NP=256, KP=128; const int wgs=256;
  float *V,*W,*H,*G,*slab; unsigned long long* dbg;
  CK(hipMalloc(&V,mp*NP*4)); CK(hipMalloc(&W,mp*KP*4)); CK(hipMalloc(&H,KP*NP*4)); CK(hipMalloc(&G,KP*KP*4));
  CK(hipMalloc(&slab,(size_t)wgs*KP*(NP+KP)*4)); CK(hipMalloc(&dbg,wgs*4*9*8));
  fillk<<<(mp*NP+255)/256,256>>>(V,mp*NP,1); fillk<<<(mp*KP+255)/256,256>>>(W,mp*KP,2);
  fillk<<<(KP*NP+255)/256,256>>>(H,KP*NP,3); fillk<<<(KP*KP+255)/256,256>>>(G,KP*KP,4);
  size_t smem=coop_smem_bytes<2,4,4>();
  CK(hipFuncSetAttribute((const void*)&k_nmf_coop<2,4,4,0>, hipFuncAttributeMaxDynamicSharedMemorySize,(int)smem));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int ntiles=(int)(mp/64);
  for(int it=0; it<4; ++it){
    hipEventRecord(e0);
    k_nmf_coop<2,4,4,0><<<wgs,256,smem>>>(V,W,H,G,ntiles/wgs,ntiles%wgs,0.f,slab,nullptr,dbg);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms,e0,e1);
    std::vector<unsigned long long> h(wgs*4*9); CK(hipMemcpy(h.data(),dbg,h.size()*8,hipMemcpyDeviceToHost));
    double s[8]={0,0,0,0,0,0,0,0}; double nt=0;
    for(int w=0; w<wgs*4; ++w){ for(int q=0;q<8;++q) s[q]+=h[w*9+q]; nt+=h[w*9+8]; }
    double tot=0; for(int q=0;q<8;++q) tot+=s[q];
    printf("it %d: %.3f ms; per tile cycles: dma-wait+barrier %.0f | phase A %.0f (ideal 24576) | epilogue %.0f | P %.0f (ideal 16384) | barrier %.0f | S+dma %.0f (ideal 8192) | total %.0f (ideal 49152)\n",
           it, ms, s[0]/nt,s[1]/nt,s[2]/nt,s[3]/nt,s[4]/nt,s[5]/nt,tot/nt);
  }
  return 0;
}
