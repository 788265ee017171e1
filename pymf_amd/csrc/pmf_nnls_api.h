// pmf_nnls_api.h -- host-side entry points of the NMFALS / NMFNNLS sub-problem kernels (pymf/nmfals.py:70-97).
//
// The kernels behind them (pmf_nnls.h, pmf_nnls_quad.h, pmf_nnls_wave.h: factorisations written out as straight-line code by
// static_for -- by far the slowest templates of the library to compile) live in a translation unit of their own,
// pmf_nnls_tu.hip, which pymf_amd/csrc/build.py compiles beside pmf_api.hip: the library builds in the time of its slower
// half.  No device code crosses the two units; pmf_api.hip sees these declarations only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// QuadCtl (all null: every problem on this launch's frame): dlist / dcount -- the problems QN = 16 leaves to the QN = 32 launch
// behind it, as a compact list (that launch returns before staging anything when the list is empty); nbig_* -- how many
// problems of a half step START beyond 16 unknowns, this call's count, the previous call's (read) and the next call's
// (zeroed here: no memset launches between the kernels).
struct QuadCtl {
  int* dlist;
  int* dcount;            // this call's list length
  int* dcount_next;       // the next call's: zeroed by the first launch of this call
  int* nbig;              // this call's count
  const int* nbig_prev;   // the previous call's
  int* nbig_next;         // the next call's: zeroed likewise
  unsigned long long* stats;   // or NULL: [2 frames][4] running totals of this site -- wave tasks, passes, sum over the passes of the
                               // largest system among the wave's four problems (what the frame-padded elimination runs over), problems
                               // solved here (bench.py's roofline block reads them live: pmf_nnqp_counters); one atomic per wave
};

// F(var, prob) = F[var * f_sk + prob * f_sp]; X likewise.  Return PMF_OK or a PMF_E* status; launches go to stream s.
int pmf_launch_nnqp(hipStream_t s, int KP, int k, const double* Hd, const float* F, int64_t f_sk, int64_t f_sp, float* X, int64_t x_sk,
                    int64_t x_sp, int64_t nprob, const int* warm, double* scratch, int skip_if_warm);
int pmf_launch_nnqp_quad(hipStream_t s, int KP, int k, const double* Horig, const double* Hd, const double* Bd, const float* F, int64_t f_sk,
                         int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm, const QuadCtl* ctl, bool count);
int pmf_launch_nnqp_wave(hipStream_t s, int KP, int k, const double* Horig, const double* Hd, const double* Bd, const float* F, int64_t f_sk,
                         int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm, double* Y0);
void pmf_launch_spd_unique_big(hipStream_t s, const double* Hd, int KP, int k, double* M, int* flag);
void pmf_launch_hessian_from_ps(hipStream_t s, const float* PS, int64_t ldp, int np, int KP, int k, double* Gd);
int pmf_nnqp_big_vpl(int k);                         // variables per lane of k_nnqp_big at k bases
int64_t pmf_nnqp_big_blocks(int k, int64_t nprob);   // workgroups (= inverse images) it is launched with
