// diagnostic: where a workgroup of k_rowgemm<4,EPI_NMF_W> spends its cycles (cfg4 shape)
#define PMF_RG_STAMPS
#include <hip/hip_runtime.h>
#include "/root/repo/pymf_amd/csrc/pmf_dev.h"
#include "/root/repo/pymf_amd/csrc/pmf_tiled.h"
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void fillk(float* p, size_t n, unsigned seed){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; if(i<n) p[i]=u01_from(seed,i); }
__global__ void k_tot(unsigned long long* out){ }
#ifndef ST_NP
#define ST_NP 256
#define ST_MP 1048576
#endif
int main(){
  constexpr int NT=4, KP=64, NP=ST_NP; const int64_t mp=ST_MP;
  float *V,*W,*H,*G;
  CK(hipMalloc(&V,mp*NP*4)); CK(hipMalloc(&W,mp*KP*4)); CK(hipMalloc(&H,KP*NP*4)); CK(hipMalloc(&G,KP*KP*4));
  fillk<<<(mp*NP+255)/256,256>>>(V,mp*NP,1); fillk<<<(mp*KP+255)/256,256>>>(W,mp*KP,2);
  fillk<<<(KP*NP+255)/256,256>>>(H,KP*NP,3); fillk<<<(KP*KP+255)/256,256>>>(G,KP*KP,4);
  const size_t smem = rowgemm_smem_bytes<NT>();
  CK(hipFuncSetAttribute((const void*)&k_rowgemm<NT,EPI_NMF_W>, hipFuncAttributeMaxDynamicSharedMemorySize,(int)smem));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it=0; it<3; ++it){
    unsigned long long z[4]={0,0,0,0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rg_acc), z, sizeof(z)));
    hipEventRecord(e0);
    k_rowgemm<NT,EPI_NMF_W><<<(unsigned)(mp/64/8),256,smem>>>(V,(int64_t)NP,NP,H,(int64_t)NP,W,G,nullptr,(int64_t)KP,0.f,mp,KP,(int)(mp/64),8);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms,e0,e1);
    CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_rg_acc), sizeof(z)));
    printf("%.3f ms; workgroup 100, all its tiles (8 tiles x (NP/64 + 1) panels): store %llu barrier %llu issue-loads %llu reads+MFMA %llu cycles (ideal MFMA 6 x 2048)\n", ms, z[0], z[1], z[2], z[3]);
  }
  return 0;
}
