// pmf_nnls.h -- batched non-negative QP for NMFALS (pymf/nmfals.py:70-97).
//
// Every column of H (resp. row of W) solves
//     minimise 1/2 x' HA x - f' x   subject to x >= 0,      HA = W^T W (resp. H H^T)
// which is the QP the reference hands to cvxopt.solvers.qp(HA, FA = -f, -I, 0)
// (nmfals.py:74,89).  HA is shared by all problems of a half step, only f differs.
//
// X holds the previous iterate on entry; when *warm_flag != 0 its support seeds the passive set.
// The flag (k_inverse_spd_mfma's `spd_flag`) is set iff HA, apart from dead (zero) rows, is well conditioned -- then the
// minimiser is unique and the starting point cannot change it; a rank-deficient HA (the reference
// test's rank-3 data with 4 bases) keeps the cold start, the path the reference's solver walks.
// One wave per problem, lane t <-> variable t (k <= 64), everything in float64 as the
// reference forces (nmfals.py:73,78).  Exact active set (Lawson-Hanson on the Gram
// matrix): the inverse of HA restricted to the passive set is kept explicitly, row t in
// lane t's registers (A[c] = inv[t][c]), and is bordered / down-dated by rank-1 updates
// when a variable enters / leaves -- O(|P|) wave steps per change, no triangular solves,
// no cross-lane reductions except one arg-max, one sum and one min per outer iteration.
// HA sits read-only in LDS; its symmetry makes every access a conflict-free row read.
#pragma once
#include <algorithm>
#include <type_traits>
#include "pmf_dev.h"
#include "../../include/pymf_hip.h"

#ifndef PMF_NNLS_TEMPLATES_ONLY   // (non-template kernels: defined by the one translation unit that launches them)
// HA (float64, padded: identity on rows/cols >= k) from the reduced (P | S) buffer: HA = S.
__global__ void k_hessian_from_ps(const float* __restrict__ PS, int64_t ldp, int np, int KP, int k,
                                  double* __restrict__ Gd) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= KP * KP) return;
  const int r = q / KP, c = q % KP;
  double v = (double)PS[(int64_t)r * ldp + np + c];
  if (r >= k || c >= k) v = (r == c) ? 1.0 : 0.0;
  Gd[q] = v;
}

#endif

// (The uniqueness test -- flag = 1 iff the unpivoted LDL^T of HA, dead variables left out, keeps every pivot above 1e-8 of its
// diagonal entry -- is a by-product of k_inverse_spd_mfma (pmf_inv.h: `spd_flag`) since round 4; the one-wave kernel of rounds
// 2-3, k_spd_unique, took 28 us per half step.  Beyond 64 bases: k_spd_unique_big below.)

#ifdef PMF_NNQP_COUNT   // diagnostic build only (tools/nnqp_probe.hip)
__device__ unsigned long long g_nnqp_cnt[4];   // outer iterations, removals, problems, rejected borders
#define PMF_NNQP_TICK(q) do { if (t == 0) atomicAdd(&g_nnqp_cnt[q], 1ull); } while (0)
#else
#define PMF_NNQP_TICK(q) do { } while (0)
#endif

// min / max over the wave, as wave_sum_f64: DPP butterfly inside the rows of 16, the four row results as scalars
__device__ __forceinline__ double wave_min_f64(double v) {
  v = fmin(v, dpp_mov_f64<0xB1>(v));
  v = fmin(v, dpp_mov_f64<0x4E>(v));
  v = fmin(v, dpp_mov_f64<0x141>(v));
  v = fmin(v, dpp_mov_f64<0x140>(v));
  return fmin(fmin(readlane_f64(v, 0), readlane_f64(v, 16)), fmin(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ double wave_max_f64(double v) {
  v = fmax(v, dpp_mov_f64<0xB1>(v));
  v = fmax(v, dpp_mov_f64<0x4E>(v));
  v = fmax(v, dpp_mov_f64<0x141>(v));
  v = fmax(v, dpp_mov_f64<0x140>(v));
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

template <int KR>   // variable slots per problem: 16, 32 or 64 (k <= KR)
__global__ __launch_bounds__(256) void k_nnqp(const double* __restrict__ Hd, int KP, int k,
                                              const float* __restrict__ F, int64_t f_sk, int64_t f_sp,
                                              float* __restrict__ X, int64_t x_sk, int64_t x_sp,
                                              int64_t nprob, const int* __restrict__ warm_flag, int skip_if_warm) {
  const bool warm = warm_flag != nullptr && *warm_flag != 0;
  if (skip_if_warm && warm) return;                  // k_nnqp_quad (pmf_nnls_quad.h) has taken the half step
  extern __shared__ __attribute__((aligned(16))) double sH[];   // [KR][KR]
  const int tid = threadIdx.x, t = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int q = tid; q < KR * KR; q += 256) {
    const int r = q / KR, c = q % KR;
    sH[q] = (r < k && c < k) ? Hd[(int64_t)r * KP + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  // stopping tolerance on the dual: 10 eps k max|HA| (same scale as the float64 oracle)
  double hmax = 0.0;
  if (t < k) hmax = sH[t * KR + t];
  hmax = wave_max_f64(hmax);
  const double tol = 2.220446049250313e-15 * (double)k * hmax;
  const bool active = t < k;
  const unsigned long long tbit = 1ull << t;

  for (int64_t prob = (int64_t)blockIdx.x * 4 + wv; prob < nprob; prob += (int64_t)gridDim.x * 4) {
    const double f = active ? (double)F[(int64_t)t * f_sk + prob * f_sp] : 0.0;
    PMF_NNQP_TICK(2);
    double x = 0.0, w = f;
    double A[KR];
#pragma unroll
    for (int c = 0; c < KR; ++c) A[c] = 0.0;
    unsigned long long pm = 0ull, ban = 0ull;   // passive set / numerically rejected (wave-uniform)

    // ---- border the inverse with variable j: u = A h_P, sigma = HA[j][j] - h_P' u ----
    auto border = [&](int j, double rel_min) -> bool {
      const double h = sH[j * KR + (t < KR ? t : 0)];   // HA[j][t] = HA[t][j]
      // Invariant: A[c] of lane t is zero unless both t and c are passive, and x is zero off the
      // passive set -- so every sum below runs over all KR slots without a test (the extra terms
      // are exact zeros) and the loops are straight-line code.
      // slots above the highest passive index hold zeros: the loops stop there (blocks of 8, one
      // uniform branch each) -- the warm start borders in ascending order, so its early steps are short
      const int hi = 63 - __builtin_clzll(pm | (1ull << j));
      // row j of HA as SCALAR loads (j is wave-uniform; Hd carries the same identity padding as sH): the
      // operands of this loop arrive in SGPRs through the scalar memory pipe instead of two v_readlane
      // per slot -- one VALU instruction per slot instead of three
      const double* __restrict__ hrow = Hd + (size_t)__builtin_amdgcn_readfirstlane(j) * KP;
      double u = 0.0;
#pragma unroll
      for (int cb = 0; cb < KR; cb += 8) {
        if (cb > hi) break;
#pragma unroll
        for (int c = cb; c < cb + 8; ++c) u = fma(A[c], hrow[c], u);
      }
      const double hjj = hrow[__builtin_amdgcn_readfirstlane(j)];
      const double sig = hjj - wave_sum_f64(h * u);                     // u is zero off the passive set
      if (!(sig > rel_min * hjj)) { ban |= 1ull << j; return false; }   // numerically dependent column
      // bordered inverse [A + u u'/sig, -u/sig; -u'/sig, 1/sig] = A + v v'/sig with v = (u; -1):
      // row j and column j of A are zero beforehand, so ONE rank-one update writes all four parts
      const double inv = pmf_rcp_f64(sig);     // within an ulp or two of 1 / sig: the size of the update's own rounding
      const double v = (t == j) ? -1.0 : u;                             // u is zero off the passive set
      const double vi = v * inv;
      // (v through LDS -- one broadcast read per slot instead of two v_readlane -- was measured twice
      // and is slower: 11.9 vs 6.9 ms per W step at cfg3; the LDS pipe is shared by the CU's 32 waves)
#pragma unroll
      for (int cb = 0; cb < KR; cb += 8) {
        if (cb > hi) break;
#pragma unroll
        for (int c = cb; c < cb + 8; c += 4) {
          const double b0 = readlane_f64(v, c), b1 = readlane_f64(v, c + 1), b2 = readlane_f64(v, c + 2),
                       b3 = readlane_f64(v, c + 3);
          A[c] = fma(vi, b0, A[c]); A[c + 1] = fma(vi, b1, A[c + 1]);
          A[c + 2] = fma(vi, b2, A[c + 2]); A[c + 3] = fma(vi, b3, A[c + 3]);
        }
      }
      pm |= 1ull << j;
      return true;
    };
    // ---- unconstrained optimum on the passive set, step back from the feasible x if infeasible ----
    auto inner = [&]() {
      for (int it = 0; it < k + 2; ++it) {
        const bool pin = (pm & tbit) != 0ull;
        const int hi = pm ? 63 - __builtin_clzll(pm) : -1;
        double s = 0.0;
#pragma unroll
        for (int cb = 0; cb < KR; cb += 8) {
          if (cb > hi) break;
#pragma unroll
          for (int c = cb; c < cb + 8; c += 4) {
            const double b0 = readlane_f64(f, c), b1 = readlane_f64(f, c + 1), b2 = readlane_f64(f, c + 2),
                         b3 = readlane_f64(f, c + 3);
            s = fma(A[c], b0, s); s = fma(A[c + 1], b1, s); s = fma(A[c + 2], b2, s); s = fma(A[c + 3], b3, s);
          }
        }
        if (!pin) s = 0.0;
        const bool bad = pin && !(s > 0.0);
        if (__ballot(bad) == 0ull) { x = s; break; }
        const double ratio = bad ? x / (x - s) : 1.0e300;
        double alpha = wave_min_f64(ratio);
        if (!(alpha >= 0.0)) alpha = 0.0;
        if (alpha > 1.0) alpha = 1.0;
        x = pin ? fma(alpha, s - x, x) : 0.0;
        const double xmax = wave_max_f64(x);
        const double tiny = 1e-15 * fmax(1.0, xmax);
        // leave: everything that hit zero (at least the variable that defined alpha)
        unsigned long long rm = __ballot(pin && (x <= tiny || (bad && ratio == alpha)));
        while (rm) {
          const int r = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(rm));
          rm &= rm - 1ull;
          PMF_NNQP_TICK(1);
          double colr = 0.0;                     // inv[t][r]
#pragma unroll
          for (int c = 0; c < KR; ++c)
            if (c == r) colr = A[c];
          const double arr = readlane_f64(colr, r);
          const double scale = colr * pmf_rcp_f64(arr);   // zero on lanes off the passive set
          const bool isr = (t == r);
#pragma unroll
          for (int c = 0; c < KR; ++c) {
            const double arc = readlane_f64(A[c], r);     // inv[r][c], zero for c off the passive set
            const double v = fma(-scale, arc, A[c]);
            A[c] = (isr || c == r) ? 0.0 : v;
          }
          pm &= ~(1ull << r);
          if (t == r) x = 0.0;
        }
      }
    };
    // ---- dual w = f - HA x over the passive set ----
    auto dual = [&]() {
      const int hi = pm ? 63 - __builtin_clzll(pm) : -1;
      w = f;
#pragma unroll
      for (int cb = 0; cb < KR; cb += 8) {
        if (cb > hi) break;
#pragma unroll
        for (int c = cb; c < cb + 8; c += 4) {
          const double b0 = readlane_f64(x, c), b1 = readlane_f64(x, c + 1), b2 = readlane_f64(x, c + 2),
                       b3 = readlane_f64(x, c + 3);
          const int tl = t < KR ? t : 0;
          w = fma(-sH[c * KR + tl], b0, w); w = fma(-sH[(c + 1) * KR + tl], b1, w);
          w = fma(-sH[(c + 2) * KR + tl], b2, w); w = fma(-sH[(c + 3) * KR + tl], b3, w);
        }
      }
    };

    if (warm) {
      // warm start (the ALS iterates change little from one half step to the next): the support of
      // the previous solution is bordered in directly -- no arg-max, no trial solve, no dual per
      // variable -- and the previous solution is the feasible point the first inner loop leaves from
      const float x0f = active ? X[(int64_t)t * x_sk + prob * x_sp] : 0.f;
      const double x0 = (x0f > 0.f) ? (double)x0f : 0.0;
      unsigned long long todo = __ballot(x0 > 0.0);
      while (todo) {
        const int j = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(todo));
        todo &= todo - 1ull;
        border(j, 1e-13);                         // a dead basis (zero row of HA) just stays out
      }
      ban = 0ull;
      x = (pm & tbit) ? x0 : 0.0;
      if (pm) { inner(); dual(); }
    }

    for (int outer = 0; outer < 3 * k + 3; ++outer) {
      // ---- most violated dual: j = argmax w over the active (zero) set ----
      const double mine = (active && !((pm | ban) & tbit)) ? w : -1.0e300;
      const double best = wave_max_f64(mine);
      if (!(best > tol)) break;
      const int j = (int)__builtin_ctzll(__ballot(mine == best));      // lowest index among ties
      PMF_NNQP_TICK(0);
      if (!border(j, 1e-13)) { PMF_NNQP_TICK(3); continue; }
      inner();
      dual();
    }
    if (active) X[(int64_t)t * x_sk + prob * x_sp] = (float)((pm & tbit) ? x : 0.0);
  }
}

// ---- num_bases > 64: the same active-set method with the inverse in global memory --------------------
// k_nnqp keeps row t of the passive block's inverse in lane t's registers, which ends at 64 variables.
// Beyond that one wave still owns a problem, lane t holds variables t, t + 64, ... (VPL slots), and the
// inverse A (symmetric, KS x KS, KS = 64 VPL) lives in a per-wave scratch image in global memory (L2/HBM),
// addressed A[c * KS + v] so that a sweep over c reads consecutive v across the lanes.  Vectors that
// every lane needs element by element (the border vector, f, x, a removed column) pass through one
// KS-entry LDS buffer.  Only entries inside passive x passive are ever read, so nothing is zeroed
// between problems.  Same pivoting rules, tolerances and tie-breaks as k_nnqp: the results agree
// with it where both apply (k <= 64), and the minimiser is unique whenever HA is positive definite.
// A generic path, not a fast one: every set change streams |P| * k doubles of A.
#ifndef PMF_NNLS_TEMPLATES_ONLY
__global__ __launch_bounds__(1024) void k_spd_unique_big(const double* __restrict__ Hd, int KP, int k,
                                                         double* __restrict__ M /*[KP][KP] scratch*/,
                                                         int* __restrict__ flag) {
  __shared__ double d0[1024];
  __shared__ double s_dmax;
  __shared__ int s_ok;
  const int tid = threadIdx.x;
  for (int q = tid; q < KP * KP; q += 1024) M[q] = Hd[q];
  for (int j = tid; j < k; j += 1024) d0[j] = Hd[(int64_t)j * KP + j];
  if (tid == 0) s_ok = 1;
  __syncthreads();
  if (tid == 0) {
    double dm = 0.0;
    for (int j = 0; j < k; ++j) dm = fmax(dm, d0[j]);
    s_dmax = dm;
  }
  __syncthreads();
  const double dead = 1e-12 * s_dmax;
  for (int j = 0; j < k; ++j) {
    if (!(d0[j] > dead)) continue;                        // dead basis (uniform)
    const double piv = M[(int64_t)j * KP + j];
    if (!(piv > 1e-8 * d0[j])) { if (tid == 0) s_ok = 0; break; }
    const int w = k - j;                                  // columns j .. k-1 of rows j+1 .. k-1
    __syncthreads();                                      // everyone has read the pivot
    for (int q = tid; q < (k - j - 1) * w; q += 1024) {
      const int r = j + 1 + q / w, c = j + q % w;
      if (!(d0[r] > dead)) continue;
      const double l = M[(int64_t)r * KP + j] / piv;
      // column j of row r is read by all of that row's threads and rewritten by one of them (c == j):
      // it goes last, behind the barrier below
      if (c != j) M[(int64_t)r * KP + c] = fma(-l, M[(int64_t)j * KP + c], M[(int64_t)r * KP + c]);
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) flag[0] = s_ok;
}
#endif

template <int VPL>   // variables per lane: 2, 4, 8 or 16 (k <= 64 VPL)
__global__ __launch_bounds__(64) void k_nnqp_big(const double* __restrict__ Hd, int KP, int k,
                                                 const float* __restrict__ F, int64_t f_sk, int64_t f_sp,
                                                 float* __restrict__ X, int64_t x_sk, int64_t x_sp,
                                                 int64_t nprob, const int* __restrict__ warm_flag,
                                                 double* __restrict__ scratch, int skip_if_warm) {
  constexpr int KS = 64 * VPL;
  __shared__ double vb[KS];
  const bool warm = warm_flag != nullptr && *warm_flag != 0;
  if (skip_if_warm && warm) return;                  // k_nnqp_wave (pmf_nnls_wave.h) has taken the half step
  const int t = threadIdx.x;
  double* __restrict__ A = scratch + (size_t)blockIdx.x * KS * KS;
  bool act[VPL];
  double hmax = 0.0;
#pragma unroll
  for (int s = 0; s < VPL; ++s) {
    const int v = t + 64 * s;
    act[s] = v < k;
    if (act[s]) hmax = fmax(hmax, Hd[(int64_t)v * KP + v]);
  }
  hmax = wave_max_f64(hmax);
  const double tol = 2.220446049250313e-15 * (double)k * hmax;
  const unsigned long long tbit = 1ull << t;

  for (int64_t prob = blockIdx.x; prob < nprob; prob += gridDim.x) {
    double x[VPL], w[VPL], f[VPL];
    unsigned long long pm[VPL], ban[VPL];
#pragma unroll
    for (int s = 0; s < VPL; ++s) {
      f[s] = act[s] ? (double)F[(int64_t)(t + 64 * s) * f_sk + prob * f_sp] : 0.0;
      x[s] = 0.0; w[s] = f[s]; pm[s] = 0ull; ban[s] = 0ull;
    }

    // acc[s] = sum over passive c of A[c][v] * coef(c), coef from the LDS vector or a row of HA
    auto sweep_lds = [&](double (&acc)[VPL]) {
#pragma unroll
      for (int s = 0; s < VPL; ++s) acc[s] = 0.0;
#pragma unroll
      for (int sc = 0; sc < VPL; ++sc) {
        unsigned long long mb = pm[sc];
        while (mb) {
          const int c = 64 * sc + (int)__builtin_ctzll(mb);
          mb &= mb - 1ull;
          const double bc = vb[c];
          const double* col = A + (size_t)c * KS + t;
#pragma unroll
          for (int s = 0; s < VPL; ++s) acc[s] = fma(col[64 * s], bc, acc[s]);
        }
      }
#pragma unroll
      for (int s = 0; s < VPL; ++s) if (!(pm[s] & tbit)) acc[s] = 0.0;
    };
    auto put = [&](const double (&src)[VPL]) {               // vb[v] = src of variable v
      __syncthreads();
#pragma unroll
      for (int s = 0; s < VPL; ++s) vb[t + 64 * s] = src[s];
      __syncthreads();
    };

    auto border = [&](int j, double rel_min) -> bool {
      const double* __restrict__ hrow = Hd + (size_t)j * KP;   // HA[j][.] = HA[.][j]
      double u[VPL];
#pragma unroll
      for (int s = 0; s < VPL; ++s) u[s] = 0.0;
#pragma unroll
      for (int sc = 0; sc < VPL; ++sc) {
        unsigned long long mb = pm[sc];
        while (mb) {
          const int c = 64 * sc + (int)__builtin_ctzll(mb);
          mb &= mb - 1ull;
          const double hc = hrow[c];
          const double* col = A + (size_t)c * KS + t;
#pragma unroll
          for (int s = 0; s < VPL; ++s) u[s] = fma(col[64 * s], hc, u[s]);
        }
      }
      double part = 0.0;
#pragma unroll
      for (int s = 0; s < VPL; ++s) {
        if (!(pm[s] & tbit)) u[s] = 0.0;
        else part = fma(hrow[t + 64 * s], u[s], part);
      }
      const double hjj = hrow[j];
      const double sig = hjj - wave_sum_f64(part);
      if (!(sig > rel_min * hjj)) { ban[j >> 6] |= 1ull << (j & 63); return false; }
      const double inv = pmf_rcp_f64(sig);
      double vv[VPL];
#pragma unroll
      for (int s = 0; s < VPL; ++s) vv[s] = (t + 64 * s == j) ? -1.0 : u[s];
      put(vv);
      pm[j >> 6] |= 1ull << (j & 63);
#pragma unroll
      for (int sc = 0; sc < VPL; ++sc) {
        unsigned long long mb = pm[sc];
        while (mb) {
          const int c = 64 * sc + (int)__builtin_ctzll(mb);
          mb &= mb - 1ull;
          const double bc = vb[c];
          double* col = A + (size_t)c * KS + t;
#pragma unroll
          for (int s = 0; s < VPL; ++s) {
            if (!(pm[s] & tbit)) continue;
            const bool fresh = (c == j) || (t + 64 * s == j);   // row / column j held nothing before
            const double old = fresh ? 0.0 : col[64 * s];
            col[64 * s] = fma(vv[s] * inv, bc, old);
          }
        }
      }
      return true;
    };

    auto inner = [&]() {
      for (int it = 0; it < k + 2; ++it) {
        put(f);
        double sv[VPL];
        sweep_lds(sv);
        bool bad[VPL];
        unsigned long long anybad = 0ull;
#pragma unroll
        for (int s = 0; s < VPL; ++s) {
          bad[s] = (pm[s] & tbit) && !(sv[s] > 0.0);
          anybad |= __ballot(bad[s]);
        }
        if (anybad == 0ull) {
#pragma unroll
          for (int s = 0; s < VPL; ++s) x[s] = sv[s];
          break;
        }
        double ratio[VPL], rmin = 1.0e300;
#pragma unroll
        for (int s = 0; s < VPL; ++s) { ratio[s] = bad[s] ? x[s] / (x[s] - sv[s]) : 1.0e300; rmin = fmin(rmin, ratio[s]); }
        double alpha = wave_min_f64(rmin);
        if (!(alpha >= 0.0)) alpha = 0.0;
        if (alpha > 1.0) alpha = 1.0;
        double xm = 0.0;
#pragma unroll
        for (int s = 0; s < VPL; ++s) { x[s] = (pm[s] & tbit) ? fma(alpha, sv[s] - x[s], x[s]) : 0.0; xm = fmax(xm, x[s]); }
        const double xmax = wave_max_f64(xm);
        const double tiny = 1e-15 * fmax(1.0, xmax);
        unsigned long long rm[VPL];
#pragma unroll
        for (int s = 0; s < VPL; ++s)
          rm[s] = __ballot((pm[s] & tbit) && (x[s] <= tiny || (bad[s] && ratio[s] == alpha)));
#pragma unroll
        for (int sr = 0; sr < VPL; ++sr) {
          while (rm[sr]) {
            const int r = 64 * sr + (int)__builtin_ctzll(rm[sr]);
            rm[sr] &= rm[sr] - 1ull;
            double colr[VPL];                                  // inv[v][r] = inv[r][v]
#pragma unroll
            for (int s = 0; s < VPL; ++s) colr[s] = (pm[s] & tbit) ? A[(size_t)r * KS + t + 64 * s] : 0.0;
            put(colr);
            const double rarr = pmf_rcp_f64(vb[r]);
            pm[r >> 6] &= ~(1ull << (r & 63));
#pragma unroll
            for (int sc = 0; sc < VPL; ++sc) {
              unsigned long long mb = pm[sc];
              while (mb) {
                const int c = 64 * sc + (int)__builtin_ctzll(mb);
                mb &= mb - 1ull;
                const double arc = vb[c];
                double* col = A + (size_t)c * KS + t;
#pragma unroll
                for (int s = 0; s < VPL; ++s)
                  if (pm[s] & tbit) col[64 * s] = fma(-(colr[s] * rarr), arc, col[64 * s]);
              }
            }
#pragma unroll
            for (int s = 0; s < VPL; ++s) if (t + 64 * s == r) x[s] = 0.0;
          }
        }
      }
    };

    auto dual = [&]() {
      put(x);
#pragma unroll
      for (int s = 0; s < VPL; ++s) w[s] = f[s];
#pragma unroll
      for (int sc = 0; sc < VPL; ++sc) {
        unsigned long long mb = pm[sc];
        while (mb) {
          const int c = 64 * sc + (int)__builtin_ctzll(mb);
          mb &= mb - 1ull;
          const double xc = vb[c];
          const double* hcol = Hd + (size_t)c * KP + t;        // HA[c][v] = HA[v][c]
#pragma unroll
          for (int s = 0; s < VPL; ++s) if (act[s]) w[s] = fma(-hcol[64 * s], xc, w[s]);
        }
      }
    };

    bool any_passive = false;
    if (warm) {
      double x0[VPL];
      unsigned long long todo[VPL];
#pragma unroll
      for (int s = 0; s < VPL; ++s) {
        const float x0f = act[s] ? X[(int64_t)(t + 64 * s) * x_sk + prob * x_sp] : 0.f;
        x0[s] = (x0f > 0.f) ? (double)x0f : 0.0;
        todo[s] = __ballot(x0[s] > 0.0);
      }
#pragma unroll
      for (int sj = 0; sj < VPL; ++sj) {
        while (todo[sj]) {
          const int j = 64 * sj + (int)__builtin_ctzll(todo[sj]);
          todo[sj] &= todo[sj] - 1ull;
          border(j, 1e-13);
        }
      }
#pragma unroll
      for (int s = 0; s < VPL; ++s) { ban[s] = 0ull; x[s] = (pm[s] & tbit) ? x0[s] : 0.0; any_passive |= pm[s] != 0ull; }
      if (any_passive) { inner(); dual(); }
    }

    for (int outer = 0; outer < 3 * k + 3; ++outer) {
      double mine[VPL], mx = -1.0e300;
#pragma unroll
      for (int s = 0; s < VPL; ++s) {
        mine[s] = (act[s] && !((pm[s] | ban[s]) & tbit)) ? w[s] : -1.0e300;
        mx = fmax(mx, mine[s]);
      }
      const double best = wave_max_f64(mx);
      if (!(best > tol)) break;
      int j = -1;
#pragma unroll
      for (int s = 0; s < VPL; ++s) {                          // lowest index among ties
        const unsigned long long eq = __ballot(mine[s] == best);
        if (j < 0 && eq) j = 64 * s + (int)__builtin_ctzll(eq);
      }
      if (!border(j, 1e-13)) continue;
      inner();
      dual();
    }
#pragma unroll
    for (int s = 0; s < VPL; ++s)
      if (act[s]) X[(int64_t)(t + 64 * s) * x_sk + prob * x_sp] = (float)((pm[s] & tbit) ? x[s] : 0.0);
    __syncthreads();
  }
}

// scratch doubles k_nnqp_big needs for `blocks` workgroups
#ifndef PMF_NNLS_TEMPLATES_ONLY   // (a launcher's body instantiates its kernels wherever it is parsed)
static inline int nnqp_big_vpl(int k) { return k <= 128 ? 2 : k <= 256 ? 4 : k <= 512 ? 8 : 16; }
static inline int64_t nnqp_big_blocks(int k, int64_t nprob) {
  const int64_t KS = 64 * nnqp_big_vpl(k);
  int64_t blocks = std::min<int64_t>(nprob, 4096);
  const int64_t cap = ((int64_t)2 << 30) / (KS * KS * 8);      // at most 2 GiB of inverse images
  if (blocks > cap) blocks = cap;
  return blocks < 1 ? 1 : blocks;
}

// F(var, prob) = F[var * f_sk + prob * f_sp]; X likewise.  Hd: [KP][KP] float64.
static inline int launch_nnqp(hipStream_t s, int KP, int k, const double* Hd, const float* F, int64_t f_sk,
                              int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm = nullptr,
                              double* scratch = nullptr, int skip_if_warm = 0) {
  if (k > 64) {                 // generic kernel; scratch: nnqp_big_blocks(k, nprob) * (64 VPL)^2 doubles
    if (!scratch || k > 1024) return PMF_EINVAL;
    const unsigned blocks = (unsigned)nnqp_big_blocks(k, nprob);
    switch (nnqp_big_vpl(k)) {
      case 2: hipLaunchKernelGGL((k_nnqp_big<2>), dim3(blocks), dim3(64), 0, s, Hd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, scratch, skip_if_warm); break;
      case 4: hipLaunchKernelGGL((k_nnqp_big<4>), dim3(blocks), dim3(64), 0, s, Hd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, scratch, skip_if_warm); break;
      case 8: hipLaunchKernelGGL((k_nnqp_big<8>), dim3(blocks), dim3(64), 0, s, Hd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, scratch, skip_if_warm); break;
      default: hipLaunchKernelGGL((k_nnqp_big<16>), dim3(blocks), dim3(64), 0, s, Hd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, scratch, skip_if_warm); break;
    }
    return PMF_OK;
  }
  const int KR = k <= 16 ? 16 : k <= 32 ? 32 : 64;
  const size_t smem = (size_t)KR * KR * sizeof(double);
  int64_t blocks = (nprob + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  switch (KR) {
    case 16: hipLaunchKernelGGL((k_nnqp<16>), dim3((unsigned)blocks), dim3(256), smem, s, Hd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, skip_if_warm); break;
    case 32: hipLaunchKernelGGL((k_nnqp<32>), dim3((unsigned)blocks), dim3(256), smem, s, Hd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, skip_if_warm); break;
    default: hipLaunchKernelGGL((k_nnqp<64>), dim3((unsigned)blocks), dim3(256), smem, s, Hd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, skip_if_warm); break;
  }
  return PMF_OK;
}
#endif
