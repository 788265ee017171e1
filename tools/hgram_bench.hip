// diagnostic: launch time of k_nmf_h_gram<NT,false> (H step + Gram; NT = 4: cfg4, -DHG_NT=8: 128 bases) for several grid sizes
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pymf_amd/csrc [-DHG_NT=8] [-DHG_OLD (round-5 sources: two template arguments)] tools/hgram_bench.hip
#include <hip/hip_runtime.h>
#include "pmf_dev.h"
#include "pmf_small.h"
#ifndef HG_NT
#define HG_NT 4
#endif
#ifdef HG_OLD
#define HG_KERNEL k_nmf_h_gram<NT,false>
#else
#define HG_KERNEL k_nmf_h_gram<NT,false,false>
#endif
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void fillk(float* p, size_t n, unsigned seed){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; if(i<n) p[i]=u01_from(seed,i); }
int main(int argc, char** argv){
  constexpr int NT=HG_NT, KP=16*NT;
  const int NP = argc > 1 ? atoi(argv[1]) : 256;
  float *H,*PS,*G,*Gpart; double *Gd,*tout,*t1p; unsigned* ticket;
  CK(hipMalloc(&H,KP*NP*4)); CK(hipMalloc(&PS,KP*(NP+KP)*4)); CK(hipMalloc(&G,KP*KP*4)); CK(hipMalloc(&Gd,KP*KP*8)); CK(hipMalloc(&tout,16));
  CK(hipMalloc(&Gpart,64*KP*KP*4)); CK(hipMalloc(&t1p,64*16)); CK(hipMalloc(&ticket,4)); CK(hipMemset(ticket,0,4));
  fillk<<<(KP*NP+255)/256,256>>>(H,KP*NP,3); fillk<<<(KP*(NP+KP)+255)/256,256>>>(PS,KP*(NP+KP),5);
  constexpr size_t smem = hgram_smem_bytes<NT>();
  CK(hipFuncSetAttribute((const void*)&HG_KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize,(int)smem));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs = 1; wgs <= NP/64; wgs *= 2){
    for (int rep=0; rep<2; ++rep){
      hipEventRecord(e0);
      for(int it=0; it<200; ++it) HG_KERNEL<<<wgs,1024,smem>>>(H,NP,PS,G,(argc>2)?nullptr:Gd,0.f,(it&1)?tout:nullptr,Gpart,t1p,ticket,nullptr,(argc>3)?0:1,IpcPeers{},0u,0,nullptr,0ull,nullptr);
      hipEventRecord(e1); CK(hipDeviceSynchronize());
      float ms; hipEventElapsedTime(&ms,e0,e1);
      if (rep) printf("np=%d wgs=%d: %.2f us/launch back to back\n", NP, wgs, ms*1000/200);
    }
  }
  return 0;
}
