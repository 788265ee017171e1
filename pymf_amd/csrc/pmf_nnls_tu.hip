// pmf_nnls_tu.hip -- the NMFALS / NMFNNLS sub-problem kernels (pymf/nmfals.py:70-97) as a translation unit of their own
// (see pmf_nnls_api.h): compiled beside pmf_api.hip by pymf_amd/csrc/build.py, linked into the same libpymf_hip.so.
#include <hip/hip_runtime.h>
#include "pmf_dev.h"
#include "pmf_nnls.h"
#include "pmf_nnls_quad.h"
#include "pmf_nnls_wave.h"
#include "pmf_nnls_api.h"

int pmf_launch_nnqp(hipStream_t s, int KP, int k, const double* Hd, const float* F, int64_t f_sk, int64_t f_sp, float* X, int64_t x_sk,
                    int64_t x_sp, int64_t nprob, const int* warm, double* scratch, int skip_if_warm) {
  return launch_nnqp(s, KP, k, Hd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, scratch, skip_if_warm);
}
int pmf_launch_nnqp_quad(hipStream_t s, int KP, int k, const double* Horig, const double* Hd, const double* Bd, const float* F, int64_t f_sk,
                         int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm, const QuadCtl* ctl, bool count) {
  return launch_nnqp_quad(s, KP, k, Horig, Hd, Bd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, ctl, count);
}
int pmf_launch_nnqp_wave(hipStream_t s, int KP, int k, const double* Horig, const double* Hd, const double* Bd, const float* F, int64_t f_sk,
                         int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm, double* Y0) {
  return launch_nnqp_wave(s, KP, k, Horig, Hd, Bd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, Y0);
}
void pmf_launch_spd_unique_big(hipStream_t s, const double* Hd, int KP, int k, double* M, int* flag) {
  hipLaunchKernelGGL(k_spd_unique_big, dim3(1), dim3(1024), 0, s, Hd, KP, k, M, flag);
}
void pmf_launch_hessian_from_ps(hipStream_t s, const float* PS, int64_t ldp, int np, int KP, int k, double* Gd) {
  hipLaunchKernelGGL(k_hessian_from_ps, dim3((unsigned)((KP * KP + 255) / 256)), dim3(256), 0, s, PS, ldp, np, KP, k, Gd);
}
int pmf_nnqp_big_vpl(int k) { return nnqp_big_vpl(k); }
int64_t pmf_nnqp_big_blocks(int k, int64_t nprob) { return nnqp_big_blocks(k, nprob); }

#ifdef PMF_QUAD_COUNT   // diagnostic build only (tools/quad_counts.py)
extern "C" int pmf_debug_quad_counts(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_quad_cnt), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_quad_t), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_quad_cnt), z, sizeof(z)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_quad_t), z, 8 * sizeof(unsigned long long)); }
  return 0;
}
#endif
