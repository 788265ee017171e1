// diagnostic: per-section cycle shares of k_nmf_fused<4,4,STAMP_MODE> (cfg4)
#define PMF_STAMPS
#ifndef STAMP_MODE
#define STAMP_MODE 0
#endif
#ifndef ST_NT
#define ST_NT 4
#define ST_NPANEL 4
#define ST_SPLIT 1
#define ST_NP 256
#endif
#define PMF_FUSED_KERNEL_ONLY
#include "/root/repo/pymf_amd/csrc/pmf_fused.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void fillk(float* p, size_t n, unsigned seed){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; if(i<n) p[i]=u01_from(seed,i); }
int main(int argc, char** argv){
  const int64_t mp = argc > 1 ? atoll(argv[1]) : 1048576; const int NP=ST_NP, KP=16*ST_NT; const int wgs=256;
  float *V,*W,*H,*G,*slab; unsigned long long* dbg;
  CK(hipMalloc(&V,mp*NP*4)); CK(hipMalloc(&W,mp*KP*4)); CK(hipMalloc(&H,KP*NP*4)); CK(hipMalloc(&G,KP*KP*4));
  CK(hipMalloc(&slab,(size_t)wgs*KP*(NP+KP)*4)); CK(hipMalloc(&dbg,wgs*4*8*8));
  fillk<<<(mp*NP+255)/256,256>>>(V,mp*NP,1); fillk<<<(mp*KP+255)/256,256>>>(W,mp*KP,2);
  fillk<<<(KP*NP+255)/256,256>>>(H,KP*NP,3); fillk<<<(KP*KP+255)/256,256>>>(G,KP*KP,4);
  size_t smem=fused_smem_bytes<ST_NT,ST_NPANEL,ST_SPLIT>();
  CK(hipFuncSetAttribute((const void*)&k_nmf_fused<ST_NT,ST_NPANEL,STAMP_MODE,ST_SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize,(int)smem));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for(int it=0; it<5; ++it){
    hipEventRecord(e0);
    k_nmf_fused<ST_NT,ST_NPANEL,STAMP_MODE,ST_SPLIT><<<wgs,256,smem>>>(V,W,H,G,(int)(mp/16/(wgs*(ST_SPLIT==2?2:4))),(int)((mp/16)%(wgs*(ST_SPLIT==2?2:4))),0.f,slab,FusedCtl{nullptr,nullptr,nullptr,0.0,0.0,0.0,0,-1},0,dbg);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms,e0,e1);
    std::vector<unsigned long long> h(wgs*4*8); CK(hipMemcpy(h.data(),dbg,h.size()*8,hipMemcpyDeviceToHost));
    double s[5]={0,0,0,0,0}; double nb=0; double pro=0, tail0=0, tailmax=0;
    for(int w=0; w<wgs*4; ++w){ for(int q=0;q<5;++q) s[q]+=h[w*8+q]; nb+=h[w*8+5]; pro+=h[w*8+6]; if(w%4==0) tail0+=h[w*8+7]; if(h[w*8+7]>tailmax) tailmax=h[w*8+7]; }
    printf("   prologue %.0f cycles (mean/wave), tail wave0 mean %.0f, tail max %.0f\n", pro/(wgs*4), tail0/wgs, tailmax);
    double tot=s[0]+s[1]+s[2]+s[3]+s[4];
    printf("it %d: %.3f ms; per block cycles: wait %.0f phaseA %.0f dmaW %.0f epi+S %.0f phaseB %.0f total %.0f (stamp units)\n", it, ms, s[0]/nb,s[1]/nb,s[2]/nb,s[3]/nb,s[4]/nb,tot/nb);
  }
  return 0;
}
