// pmf_nnls.h -- batched non-negative QP solver for NMFALS (placeholder until built).
#pragma once
#include "pmf_dev.h"
#include "../../include/pymf_hip.h"
__global__ void k_hessian_from_ps(const float* __restrict__ PS, int64_t ldp, int np, int KP, int k, double* __restrict__ Gd) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= KP * KP) return;
  const int r = q / KP, c = q % KP;
  double v = (double)PS[(int64_t)r * ldp + np + c];
  if (r >= k || c >= k) v = (r == c) ? 1.0 : 0.0;
  Gd[q] = v;
}
static inline int launch_nnqp(hipStream_t, int, int, const double*, const float*, int64_t, int64_t, float*, int64_t, int64_t, int64_t) { return PMF_EINVAL; }
