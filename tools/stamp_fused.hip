// diagnostic: per-section cycle shares of k_nmf_fused<ST_NT,ST_NPANEL,STAMP_MODE,ST_SPLIT>, and (round 5) where a LAUNCH's time goes:
// in-kernel span on the 100 MHz wall clock (first wave's start .. last wave's end) against the HIP-event time of the launch,
// the shader clock the waves really ran at (s_memtime cycles / wall-clock time), start skew of the workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 [-DST_NT=2 -DST_NPANEL=4 -DST_SPLIT=2 -DST_NP=512] tools/stamp_fused.hip -o ...
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
#include "../pymf_amd/csrc/pmf_fused.h"
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__global__ void fillk(float* p, size_t n, unsigned seed){ size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; if(i<n) p[i]=u01_from(seed,i); }
__global__ void spacer(unsigned long long ticks){ const unsigned long long t0=wall_clock64(); while(wall_clock64()-t0<ticks) __builtin_amdgcn_s_sleep(8); }
int main(int argc, char** argv){
  const int64_t mp = argc > 1 ? atoll(argv[1]) : 1048576; const int NP=ST_NP, KP=16*ST_NT;
  const int per = ST_SPLIT==2 ? 2 : 4;
  const int wgs = (int)std::min<int64_t>(256, (mp/16+per-1)/per);
  float *V,*W,*H,*G,*slab; unsigned long long* dbg;
  CK(hipMalloc(&V,mp*NP*4)); CK(hipMalloc(&W,mp*KP*4)); CK(hipMalloc(&H,KP*NP*4)); CK(hipMalloc(&G,KP*KP*4));
  CK(hipMalloc(&slab,(size_t)wgs*KP*(NP+KP)*4)); CK(hipMalloc(&dbg,wgs*4*12*8));
  fillk<<<(mp*NP+255)/256,256>>>(V,mp*NP,1); fillk<<<(mp*KP+255)/256,256>>>(W,mp*KP,2);
  fillk<<<(KP*NP+255)/256,256>>>(H,KP*NP,3); fillk<<<(KP*KP+255)/256,256>>>(G,KP*KP,4);
  size_t smem=fused_smem_bytes<ST_NT,ST_NPANEL,ST_SPLIT>();
  CK(hipFuncSetAttribute((const void*)&k_nmf_fused<ST_NT,ST_NPANEL,STAMP_MODE,ST_SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize,(int)smem));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nblk=(int)(mp/16), nw=wgs*per;
  auto run=[&](){ k_nmf_fused<ST_NT,ST_NPANEL,STAMP_MODE,ST_SPLIT><<<wgs,256,smem>>>(V,W,H,G,nblk/nw,nblk%nw,0.f,slab,FusedCtl{nullptr,nullptr,nullptr,0.0,0.0,0.0,0,-1},0,dbg); };
  // spacer: a kernel of `swg` workgroups that does nothing for `sus` microseconds (the small k x n kernels of an iteration: the chip
  // mostly idle between two one-pass launches) -- does the clock the one-pass kernel runs at depend on it?
  const int warm = argc>2?atoi(argv[2]):40; const int sus = argc>3?atoi(argv[3]):0; const int swg = argc>4?atoi(argv[4]):8;
  for(int it=0; it<5; ++it){
    for(int q=0; q<warm; ++q){ run(); if(sus) spacer<<<swg,1024>>>((unsigned long long)sus*100ull); }   // every measurement behind a hot loop
    hipEventRecord(e0);
    run();
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms,e0,e1);
    std::vector<unsigned long long> h(wgs*4*12); CK(hipMemcpy(h.data(),dbg,h.size()*8,hipMemcpyDeviceToHost));
    double s[5]={0,0,0,0,0}; double nb=0; double pro=0, tail0=0, tailmax=0, loopc=0, totc=0;
    unsigned long long rt_lo=~0ull, rt_hi=0, rt_start_hi=0, rt_end_lo=~0ull; double clk=0; int nwv=0;
    for(int w=0; w<wgs*4; ++w){ const unsigned long long* d=&h[w*12];
      for(int q=0;q<5;++q) s[q]+=d[q]; nb+=d[5]; pro+=d[6]; if(w%4==0) tail0+=d[7]; if(d[7]>tailmax) tailmax=d[7];
      rt_lo=std::min(rt_lo,d[8]); rt_hi=std::max(rt_hi,d[9]); rt_start_hi=std::max(rt_start_hi,d[8]); rt_end_lo=std::min(rt_end_lo,d[9]);
      totc+=d[10]; loopc+=d[11]; if(d[9]>d[8]){ clk+=(double)d[10]/((double)(d[9]-d[8])*10.0); ++nwv; } }
    printf("it %d: HIP events %.2f us; in-kernel span %.2f us (first start .. last end, 100 MHz clock); start skew %.2f us, end skew %.2f us; "
           "mean cycles/wave %.0f (loop %.0f) at %.3f GHz\n", it, ms*1e3, (rt_hi-rt_lo)*0.01, (rt_start_hi-rt_lo)*0.01, (rt_hi-rt_end_lo)*0.01,
           totc/(wgs*4), loopc/(wgs*4), clk/nwv);
    printf("   prologue %.0f cycles (mean/wave), tail wave0 mean %.0f, tail max %.0f\n", pro/(wgs*4), tail0/wgs, tailmax);
    double tot=s[0]+s[1]+s[2]+s[3]+s[4];
    printf("   per block cycles: wait %.0f phaseA %.0f dmaW %.0f epi+S %.0f phaseB %.0f total %.0f (blocks/wave %.2f)\n", s[0]/nb,s[1]/nb,s[2]/nb,s[3]/nb,s[4]/nb,tot/nb, nb/(wgs*4));
  }
  return 0;
}
