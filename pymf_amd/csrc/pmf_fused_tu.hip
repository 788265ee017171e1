// pmf_fused_tu.hip -- the one-pass kernels (pmf_fused.h, pmf_coop.h) as a translation unit of their own (pmf_fused_api.h).
#include <hip/hip_runtime.h>
#include "pmf_dev.h"
#include "pmf_fused.h"
#include "pmf_coop.h"
#include "pmf_fused_api.h"

int pmf_fused_grid_for(int NT, int np, int64_t mp, bool allow_split) { return fused_grid_for(NT, np, mp, allow_split); }
const char* pmf_fused_kernel_name(int NT, int np, int mode) { return fused_kernel_name(NT, np, mode); }
int pmf_launch_fused(hipStream_t s, int mode, int NT, int np, const float* V, float* W, const float* H, const float* G, int64_t mp,
                     int wgs, float lamb, float* slab, const FusedCtl& ctl, int ngp, hipEvent_t e0, hipEvent_t e1) {
  return launch_fused(s, mode, NT, np, V, W, H, G, mp, wgs, lamb, slab, ctl, ngp, e0, e1);
}
bool pmf_coop_shape(int NT, int np, int* bt, int* rb, int* npanel) { return coop_shape(NT, np, bt, rb, npanel); }
int pmf_coop_pad_np(int NT, int np) { return coop_pad_np(NT, np); }
int pmf_coop_grid_for(int64_t mp, int rb) { return coop_grid_for(mp, rb); }
int pmf_launch_coop(hipStream_t s, int mode, int NT, int np, const float* V, float* W, const float* H, const float* G, int64_t mp,
                    int wgs, float lamb, float* slab, const int* stop) {
  return launch_coop(s, mode, NT, np, V, W, H, G, mp, wgs, lamb, slab, stop);
}
void pmf_launch_reduce_slabs_coop(hipStream_t s, const float* slab, int nslabs, int bt, int ntp, int ktiles, int np, float* out, const int* stop) {
  hipLaunchKernelGGL(k_reduce_slabs_coop, dim3((unsigned)(4 * bt * (ntp + ktiles))), dim3(1024), 0, s, slab, nslabs, bt, ntp, np, out, stop);
}
