// pmf_fused.h -- fused one-pass NMF iteration kernel (placeholder until built).
#pragma once
#include "pmf_dev.h"
#include "../../include/pymf_hip.h"
static inline int fused_grid_for(int NT, int np, int64_t mp) { (void)NT; (void)np; (void)mp; return 0; }
static inline const char* fused_kernel_name(int NT, int np) { (void)NT; (void)np; return "none"; }
static inline int launch_fused(hipStream_t, int, int, const float*, float*, const float*, const float*, int64_t, int, float*) { return PMF_EINVAL; }
