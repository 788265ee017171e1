// pmf_nnls_quad_tu.hip -- k_nnqp_quad (pmf_nnls_quad.h: sixteen lanes per NMFALS sub-problem, the slowest template of the
// library to compile) as a translation unit of its own (pmf_nnls_api.h).
#include <hip/hip_runtime.h>
#define PMF_NNLS_TEMPLATES_ONLY   // k_hessian_from_ps / k_spd_unique_big belong to pmf_nnls_tu.hip
#include "pmf_dev.h"
#include "pmf_nnls_quad.h"
#include "pmf_nnls_api.h"

int pmf_launch_nnqp_quad(hipStream_t s, int KP, int k, const double* Horig, const double* Hd, const double* Bd, const float* F, int64_t f_sk,
                         int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm, const QuadCtl* ctl, bool count) {
  return launch_nnqp_quad(s, KP, k, Horig, Hd, Bd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, ctl, count);
}

#ifdef PMF_QUAD_COUNT   // diagnostic build only (tools/quad_counts.py)
extern "C" int pmf_debug_quad_counts(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_quad_cnt), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_quad_t), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_quad_cnt), z, sizeof(z)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_quad_t), z, 8 * sizeof(unsigned long long)); }
  return 0;
}
#endif
