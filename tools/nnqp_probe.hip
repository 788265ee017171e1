// diagnostic: iteration counts and time of k_nnqp<64> on an ALS-like H step (HA = W^T W, f = W^T v)
#define PMF_NNQP_COUNT
#include <hip/hip_runtime.h>
#include "/root/repo/pymf_amd/csrc/pmf_nnls.h"
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
int main(int argc, char** argv){
  const int k = 64, m = argc > 1 ? atoi(argv[1]) : 4096, n = argc > 2 ? atoi(argv[2]) : 1024;
  const double sparsity = argc > 3 ? atof(argv[3]) : 0.0;
  std::mt19937_64 g(7); std::uniform_real_distribution<double> U(0.0, 1.0);
  std::vector<double> W((size_t)m*k), HA((size_t)k*k, 0.0); std::vector<float> F((size_t)k*n), X((size_t)k*n, 0.f);
  for (auto& w : W) { w = U(g); if (U(g) < sparsity) w = 0.0; }
  for (int a=0;a<k;++a) for (int b=0;b<k;++b){ double s=0; for(int r=0;r<m;++r) s+=W[(size_t)r*k+a]*W[(size_t)r*k+b]; HA[a*k+b]=(double)(float)s; }
  std::vector<double> v(m);
  for (int c=0;c<n;++c){ for(int r=0;r<m;++r) v[r]=U(g); for(int a=0;a<k;++a){ double s=0; for(int r=0;r<m;++r) s+=W[(size_t)r*k+a]*v[r]; F[(size_t)a*n+c]=(float)s; } }
  double* dH; float *dF,*dX; CK(hipMalloc(&dH,k*k*8)); CK(hipMalloc(&dF,F.size()*4)); CK(hipMalloc(&dX,X.size()*4));
  CK(hipMemcpy(dH,HA.data(),k*k*8,hipMemcpyHostToDevice)); CK(hipMemcpy(dF,F.data(),F.size()*4,hipMemcpyHostToDevice)); CK(hipMemcpy(dX,X.data(),X.size()*4,hipMemcpyHostToDevice));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep=0; rep<2; ++rep){
    unsigned long long z[4]={0,0,0,0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_nnqp_cnt), z, sizeof(z)));
    hipEventRecord(e0);
    launch_nnqp(0, k, k, dH, dF, n, 1, dX, n, 1, n, nullptr);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms,e0,e1);
    CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_nnqp_cnt), sizeof(z)));
    printf("m=%d n=%d: %.3f ms; per problem: outer %.1f removals %.1f rejected %.2f\n", m, n, ms, (double)z[0]/z[2], (double)z[1]/z[2], (double)z[3]/z[2]);
  }
  CK(hipMemcpy(X.data(),dX,X.size()*4,hipMemcpyDeviceToHost)); int nz=0; for(float x: X) nz += x>0; printf("avg support %.1f of %d\n", (double)nz/n, k);
  return 0;
}
