// pmf_nnls_tu.hip -- the NMFALS / NMFNNLS sub-problem kernels (pymf/nmfals.py:70-97) other than k_nnqp_quad (pmf_nnls_quad_tu.hip)
// as a translation unit of their own (see pmf_nnls_api.h): compiled beside pmf_api.hip by pymf_amd/csrc/build.py.
#include <hip/hip_runtime.h>
#define PMF_QUAD_TEMPLATES_ONLY   // k_nnqp_quad and its launchers belong to pmf_nnls_quad_tu.hip
#include "pmf_dev.h"
#include "pmf_nnls.h"
#include "pmf_nnls_quad.h"
#include "pmf_nnls_wave.h"
#include "pmf_nnls_api.h"

int pmf_launch_nnqp(hipStream_t s, int KP, int k, const double* Hd, const float* F, int64_t f_sk, int64_t f_sp, float* X, int64_t x_sk,
                    int64_t x_sp, int64_t nprob, const int* warm, double* scratch, int skip_if_warm) {
  return launch_nnqp(s, KP, k, Hd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, scratch, skip_if_warm);
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
