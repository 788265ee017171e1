// lab: k_nmf_quad<2,2> (two waves per SIMD, tools/lab_fused_quad.h -- an experiment, not part of the library) against k_nmf_fused<2,4,*,SPLIT 2> on the same inputs
// (cfg2: 65 536 x 512, k = 32): W and the summed slabs compared, both kernels timed alternately.
// build: hipcc --offload-arch=gfx950 -O3 [-DPMF_STAMPS] tools/quad_lab.hip -o tools/quad_lab ; run: tools/quad_lab [rows]
#define PMF_FUSED_KERNEL_ONLY
#include "/root/repo/tools/lab_fused_quad.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
#ifndef LAB_MODE
#define LAB_MODE 0
#endif
__global__ void fillk(float* p, size_t n, unsigned seed){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; if(i<n) p[i]=u01_from(seed,i); }
int main(int argc, char** argv){
  const int64_t mp = argc > 1 ? atoll(argv[1]) : 65536; const int NP=512, KP=32; 
  int wgs = (int)std::min<int64_t>(256, (mp/16+1)/2);
  float *V,*W0,*W1,*W2,*H,*G,*slab1,*slab2; unsigned long long* dbg;
  CK(hipMalloc(&V,mp*NP*4)); CK(hipMalloc(&W0,mp*KP*4)); CK(hipMalloc(&W1,mp*KP*4)); CK(hipMalloc(&W2,mp*KP*4));
  CK(hipMalloc(&H,KP*NP*4)); CK(hipMalloc(&G,KP*KP*4));
  const size_t slab_floats = (size_t)KP*(NP+KP);   // >= NTUT*256
  CK(hipMalloc(&slab1,(size_t)wgs*slab_floats*4)); CK(hipMalloc(&slab2,(size_t)wgs*slab_floats*4)); CK(hipMalloc(&dbg,wgs*8*8*8));
  CK(hipMemset(slab1,0,(size_t)wgs*slab_floats*4)); CK(hipMemset(slab2,0,(size_t)wgs*slab_floats*4));
  fillk<<<(mp*NP+255)/256,256>>>(V,mp*NP,1); fillk<<<(mp*KP+255)/256,256>>>(W0,mp*KP,2);
  fillk<<<(KP*NP+255)/256,256>>>(H,KP*NP,3); fillk<<<(KP*KP+255)/256,256>>>(G,KP*KP,4);
  const size_t smem1=fused_smem_bytes<2,4,2>(), smem2=quad_smem_bytes<2,2>();
  CK(hipFuncSetAttribute((const void*)&k_nmf_fused<2,4,LAB_MODE,2>, hipFuncAttributeMaxDynamicSharedMemorySize,(int)smem1));
  CK(hipFuncSetAttribute((const void*)&k_nmf_quad<2,2,LAB_MODE>, hipFuncAttributeMaxDynamicSharedMemorySize,(int)smem2));
  const int nblk=(int)(mp/16), nq=wgs*2, blk_per=nblk/nq, blk_extra=nblk%nq;
  const FusedCtl ctl{nullptr,nullptr,nullptr,0.0,0.0,0.0,0,-1};
  const float lamb = 0.05f;
  auto run_old=[&](float* W,float* slab){ k_nmf_fused<2,4,LAB_MODE,2><<<wgs,256,smem1>>>(V,W,H,G,blk_per,blk_extra,lamb,slab,ctl,0
#ifdef PMF_STAMPS
    ,dbg
#endif
    ); };
  auto run_new=[&](float* W,float* slab){ k_nmf_quad<2,2,LAB_MODE><<<wgs,512,smem2>>>(V,W,H,G,blk_per,blk_extra,lamb,slab,ctl,0
#ifdef PMF_STAMPS
    ,dbg
#endif
    ); };
  // ---- results ----
  CK(hipMemcpy(W1,W0,mp*KP*4,hipMemcpyDeviceToDevice)); CK(hipMemcpy(W2,W0,mp*KP*4,hipMemcpyDeviceToDevice));
  run_old(W1,slab1); CK(hipDeviceSynchronize()); run_new(W2,slab2); CK(hipDeviceSynchronize());
  {
    std::vector<float> a((size_t)mp*KP), b((size_t)mp*KP);
    CK(hipMemcpy(a.data(),W1,a.size()*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(),W2,b.size()*4,hipMemcpyDeviceToHost));
    double worst=0, na=0; size_t bad=0; for(size_t q=0;q<a.size();++q){ double d=fabs((double)a[q]-b[q]); double r=d/(fabs((double)a[q])+1e-30); if(!(r<=worst)) worst=r; na+=a[q]; if(!(r<1e-5)) ++bad; }
    printf("W: max rel diff old vs quad %.3e, elements beyond 1e-5: %zu of %zu (mean %.4f)\n", worst, bad, a.size(), na/a.size());
    const size_t ntut = 2*32+3, per = ntut*256;
    std::vector<float> s1((size_t)wgs*per), s2((size_t)wgs*per);
    CK(hipMemcpy(s1.data(),slab1,s1.size()*4,hipMemcpyDeviceToHost)); CK(hipMemcpy(s2.data(),slab2,s2.size()*4,hipMemcpyDeviceToHost));
    double w2=0; size_t bad2=0;
    for(size_t e=0;e<per;++e){ double x=0,y=0; for(int w=0;w<wgs;++w){ x+=s1[(size_t)w*per+e]; y+=s2[(size_t)w*per+e]; } double r=fabs(x-y)/(fabs(x)+1e-30); if(!(r<=w2)) w2=r; if(!(r<1e-5)) ++bad2; }
    printf("summed slabs (P | S in tile layout): max rel diff %.3e, beyond 1e-5: %zu of %zu\n", w2, bad2, per);
  }
  // ---- timing, alternately ----
  hipEvent_t e0,e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for(int rep=0; rep<4; ++rep){
    for(int which=0; which<2; ++which){
      (void)hipEventRecord(e0);
      for(int it=0; it<50; ++it){ if(which==0) run_old(W1,slab1); else run_new(W2,slab2); }
      (void)hipEventRecord(e1); CK(hipDeviceSynchronize());
      float ms; (void)hipEventElapsedTime(&ms,e0,e1);
      printf("%s: %.2f us per launch\n", which==0?"k_nmf_fused<2,4,SPLIT 2>":"k_nmf_quad<2,2>        ", ms/50*1e3);
    }
  }
#ifdef PMF_STAMPS
  run_new(W2,slab2); CK(hipDeviceSynchronize());
  std::vector<unsigned long long> hh(wgs*8*8); CK(hipMemcpy(hh.data(),dbg,hh.size()*8,hipMemcpyDeviceToHost));
  for(int quad=0; quad<2; ++quad){
  double s[5]={0,0,0,0,0}, nb=0, pro=0, tail=0;
  for(int w=0; w<wgs*8; ++w){ if(((w%8)>>2)!=quad) continue; for(int q=0;q<5;++q) s[q]+=hh[w*8+q]; nb+=hh[w*8+5]; pro+=hh[w*8+6]; tail+=hh[w*8+7]; }
  printf("quad %d stamps: per block cycles (mean/wave): A slot %.0f, B: exchange %.0f, epilogue %.0f, phase B %.0f, barriers %.0f; prologue %.0f, tail %.0f (per block)\n",
         quad, s[0]/nb, s[1]/nb, s[2]/nb, s[3]/nb, s[4]/nb, pro/nb, tail/nb);
  }
#endif
  return 0;
}
