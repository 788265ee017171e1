// pmf_fused_api.h -- host-side entry points of the one-pass kernels (pmf_fused.h: k_nmf_fused, pmf_coop.h: k_nmf_coop; the
// update_w / update_h contractions of pymf/nmf.py:122-132 in one pass over V).
//
// Their 87 instantiations live in a translation unit of their own, pmf_fused_tu.hip, compiled beside pmf_api.hip and
// pmf_nnls_tu.hip (pymf_amd/csrc/build.py): pmf_api.hip sees these declarations only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { FUSED_NMF = 0, FUSED_SNMF = 1, FUSED_BNMF = 2, FUSED_RNMF = 3 };   // RNMF: V is D = S - data

// Free-running pmf_factorize loops.  stop: a launch enqueued behind a converged iteration is a no-op.
// conv_iter >= 0: the error and the convergence test of THAT (the previous) iteration, nmf.py:134-139,
// 198-202, are still to be evaluated from the trace terms its H step left in tt -- every workgroup does
// it for itself while its first tiles are in flight (same data, same arithmetic: the same decision
// everywhere; workgroup 0 records it), which saves the k_conv_check launch between two iterations:
// 4.8 us of a 67 us iteration at 65 536 x 512, k = 32.  Same expressions as k_conv_check (pmf_small.h).
struct FusedCtl {
  int* stop;            // [0] 0 run / 1 converged / 2 the trace identity cancels, [1] iteration; or NULL
  const double* tt;     // ntt pairs (<P,H>, <S,G>)
  double* ferr;         // device error history
  double vnorm2, eps, nsamp;
  int ntt, conv_iter;
};


// Workgroups to launch (one per CU), 0 when the shape is not covered by the k <= 64 one-pass kernel.
int pmf_fused_grid_for(int NT, int np, int64_t mp, bool allow_split);
const char* pmf_fused_kernel_name(int NT, int np, int mode);
// G: H H^T [KP][KP] float32 (NMF, BNMF, RNMF).  FUSED_SNMF: H is M^T = inv(H H^T) H and G is unused.
int pmf_launch_fused(hipStream_t s, int mode, int NT, int np, const float* V, float* W, const float* H, const float* G, int64_t mp,
                     int wgs, float lamb, float* slab, const FusedCtl& ctl, int ngp, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// the cooperative form (64 < k <= 128, or k <= 64 with 256 < n <= 512)
bool pmf_coop_shape(int NT, int np, int* bt, int* rb, int* npanel);
int pmf_coop_pad_np(int NT, int np);
int pmf_coop_grid_for(int64_t mp, int rb);
int pmf_launch_coop(hipStream_t s, int mode, int NT, int np, const float* V, float* W, const float* H, const float* G, int64_t mp,
                    int wgs, float lamb, float* slab, const int* stop);
void pmf_launch_reduce_slabs_coop(hipStream_t s, const float* slab, int nslabs, int bt, int ntp, int ktiles, int np, float* out, const int* stop);
