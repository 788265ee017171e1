// pmf_nnls_wave.h -- the NMFALS sub-problems (pymf/nmfals.py:70-97) for 64 < num_bases <= 128, ONE WAVE PER PROBLEM.
//
//     minimise 1/2 x' HA x - f' x   subject to x >= 0      (HA = H H^T or W^T W)
//
// k_nnqp_big (pmf_nnls.h) walks to the KKT point one variable at a time and keeps the explicit inverse of HA[P,P] as a
// 128 x 128 image in GLOBAL memory, built border by border: 50-130 ms for 65 536 problems at k = 128, a hundred times
// k_nnqp_quad's time at k = 64.  This kernel is k_nnqp_quad's method (pmf_nnls_quad.h: B = inv(HA) formed once per half
// step, the SMALLER of HA[P,P] / B[N,N] factorised per problem, block principal pivoting with Murty's rule as the
// safeguard -- the same rules, tolerances and pass limits) at a wave per problem:
//   * lane t owns variables t and t + 64, so the passive set is two ballots;
//   * the system has at most k / 2 <= 64 unknowns whatever the support: row i of its LDL^T factor lives in the REGISTERS
//     of lane i (64 doubles, every index a constant of the program text: straight-line code as in k_nnqp_quad); a step's
//     pivot column crosses the lanes through one 64-double LDS line (one write, broadcast reads).  A first form with the
//     factor in a 64 x 65 LDS image (left-looking, 35 KiB: one wave per SIMD) took 7.5-12 ms where this one takes
//     (profiles/r03_experiments.md);
//   * products with HA / B read rows of the shared matrices from L2 (128 KiB each: beyond LDS), coalesced.
// Preconditions as k_nnqp_quad: *warm_flag != 0 (k_inverse_spd_mfma's pivots found HA positive definite and well
// conditioned); otherwise the kernel returns at once and k_nnqp_big takes the half step.
#pragma once
#include "pmf_dev.h"
#include "pmf_nnls.h"
#include "pmf_nnls_quad.h"   // static_for

constexpr int WVN = 64;            // largest system a problem factorises (k <= 128)

// a0 += sum_p coef(p) M[row(p)][t], a1 += ... M[row(p)][t + 64] over the positions p < ns of the system's list; coef(p) is
// cv[row(p)] (BYVAR) or cv[p].  EIGHT rows -- sixteen requests -- go out together: the rows come from L2 (the matrices are
// 128 KiB each) and a wave is alone on its SIMD, so the requests in flight are the whole of its latency hiding
// (two rows at a time: 12 ms for 65 536 problems at k = 128; eight: see profiles/r03_experiments.md).
template <bool BYVAR, bool NEG>
__device__ __forceinline__ void wv_rows_dot(const double* __restrict__ M, int KP, int t, const int* lst, const double* cv, int ns,
                                            double& a0, double& a1) {
  for (int p = 0; p < ns; p += 8) {
    int c[8];
    double m[8], v0[8], v1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool ok = p + e < ns;
      c[e] = ok ? lst[p + e] : 0;
      const double mv = ok ? cv[BYVAR ? c[e] : p + e] : 0.0;
      m[e] = NEG ? -mv : mv;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double* r = M + (int64_t)c[e] * KP + t;
      v0[e] = r[0]; v1[e] = r[64];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { a0 = fma(v0[e], m[e], a0); a1 = fma(v1[e], m[e], a1); }
  }
}

__global__ __launch_bounds__(64, 2) void k_nnqp_wave(const double* __restrict__ Horig, const double* __restrict__ Hd,
                                                  const double* __restrict__ Bd, int KP, int k,
                                                  const float* __restrict__ F, int64_t f_sk, int64_t f_sp,
                                                  float* __restrict__ X, int64_t x_sk, int64_t x_sp, int64_t nprob,
                                                  const int* __restrict__ warm_flag) {
  if (*warm_flag == 0) return;
  __shared__ double vecP[2 * WVN]; // the pivot column of a factorisation step, double buffered
  __shared__ double vecV[128];     // f, then (complement form) y, by variable
  __shared__ double vecC[WVN];     // mu by position
  __shared__ int lst[WVN];         // the variable at each position of the system
  const int t = threadIdx.x;
  const unsigned long long below = (1ull << t) - 1ull;
  double hm = 0.0, dg[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int v = t + 64 * s;
    dg[s] = v < k ? Horig[(int64_t)v * KP + v] : 0.0;
    hm = fmax(hm, dg[s]);
  }
  const double hmax = wave_max_f64(hm);
  const double tol = 2.220446049250313e-15 * (double)k * hmax;       // as k_nnqp and the oracle
  unsigned long long live[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) live[s] = __ballot(t + 64 * s < k && dg[s] > 1e-12 * hmax);
  const int klive = __popcll(live[0]) + __popcll(live[1]);

  for (int64_t prob = blockIdx.x; prob < nprob; prob += gridDim.x) {
    double f[2], x[2];
    unsigned long long pm[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int v = t + 64 * s;
      const bool act = v < k;
      f[s] = act ? (double)F[(int64_t)v * f_sk + prob * f_sp] : 0.0;
      const float x0 = act ? X[(int64_t)v * x_sk + prob * x_sp] : 0.f;
      x[s] = x0 > 0.f ? (double)x0 : 0.0;
      pm[s] = __ballot(x0 > 0.f) & live[s];
    }
    bool have_y0 = false;
    double y0[2] = {0.0, 0.0};
    int npass = 0, ninf_best = k + 1, backup = 3;

    for (int pass = 0; pass < 8 * 128 + 16; ++pass) {
      const int np_ = __popcll(pm[0]) + __popcll(pm[1]), nn = klive - np_;
      const bool comp = nn < np_;                    // complement form: factorise B[N,N]
      const int ns = comp ? nn : np_;                // <= k / 2 <= 64
      unsigned long long sm[2];
      int mypos[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) sm[s] = comp ? (~pm[s] & live[s]) : pm[s];
      const int ns0 = __popcll(sm[0]);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        mypos[s] = -1;
        if ((sm[s] >> t) & 1ull) {
          mypos[s] = (s ? ns0 : 0) + __popcll(sm[s] & below);
          lst[mypos[s]] = t + 64 * s;
        }
        vecV[t + 64 * s] = f[s];
      }
      __syncthreads();
      const double* __restrict__ Msel = comp ? Bd : Hd;
      // ---- complement form: y = B (f on P) = y0 - B (f on N), y0 = B (f on the live variables) once per problem ----
      double y[2] = {0.0, 0.0};
      if (comp) {
        if (!have_y0) {
          double a0 = 0.0, a1 = 0.0;
          for (int c0 = 0; c0 < k; c0 += 16) {         // sixteen rows, 32 requests, together
            double fc[16], v0[16], v1[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int c = c0 + e;
              const bool lv = c < k && ((live[c >> 6] >> (c & 63)) & 1ull);
              fc[e] = lv ? vecV[c] : 0.0;
              const double* r = Bd + (int64_t)(c < k ? c : 0) * KP + t;
              v0[e] = r[0]; v1[e] = r[64];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) { a0 = fma(v0[e], fc[e], a0); a1 = fma(v1[e], fc[e], a1); }
          }
          y0[0] = a0; y0[1] = a1;
          have_y0 = true;
        }
        y[0] = y0[0]; y[1] = y0[1];
        wv_rows_dot<true, true>(Bd, KP, t, lst, vecV, ns, y[0], y[1]);
        __syncthreads();
        vecV[t] = y[0]; vecV[t + 64] = y[1];
        __syncthreads();
      }
      // ---- the system, right-aligned in a 64-slot frame (positions shift .. 63, identity in front: the factorisation
      //      starts at the first position in use); row i of M[S,S] in the REGISTERS of lane i, every index a constant of
      //      the program text; all 64 requests of a row go out together ----
      const int shift = WVN - ns;
      // (the lane number behind an opaque move: LLVM otherwise hoists the 64 identity entries and the 2 x 63 lane masks below
      //  out of the problem loop and spills them -- 64 doubles of scratch, 414 SGPR spills)
      int tt = t;
      asm volatile("" : "+v"(tt));
      const bool on_t = t >= shift;
      const int myvar = on_t ? lst[t - shift] : 0;
      double b = 0.0;
      if (on_t) b = comp ? -vecV[myvar] : vecV[myvar];
      double Lr[WVN];
      static_for<0, WVN>([&](auto cc_) {
        constexpr int c = decltype(cc_)::value;
        const int rv = c >= shift ? lst[c - shift] : 0;
        // (a 32-bit offset from the uniform base: one address register per request, not two -- 64 requests are in flight)
        const double val = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Msel) + (unsigned)(rv * KP + myvar) * 8u);
        Lr[c] = (on_t && c >= shift) ? val : (c == tt ? 1.0 : 0.0);
      });
      // ---- LDL^T, right-looking: step j scales column j and takes l_ij a_cj off every later column c; a_cj = A[c][j]
      //      (symmetry: lane c's own entry j) reaches all lanes through a 64-double LDS line, ONE write and broadcast reads ----
      double dv = 1.0;
      static_for<0, WVN>([&](auto jc_) {
        constexpr int j = decltype(jc_)::value;
        if (j >= shift) {
          double* col = vecP + (j & 1) * WVN;
          col[t] = Lr[j];
          __syncthreads();
          const double dj = col[j];
          const double inv = pmf_rcp_f64(dj);
          const double lij = Lr[j] * inv;
          if (tt == j) dv = dj;
          static_for<j + 1, WVN>([&](auto cc_) {
            constexpr int c = decltype(cc_)::value;
            Lr[c] = fma(-lij, col[c], Lr[c]);
          });
          Lr[j] = lij;
        }
      });
      // ---- L z = b, z / d, L^T mu = z ----
      static_for<0, WVN - 1>([&](auto jc_) {
        constexpr int j = decltype(jc_)::value;
        if (j >= shift) {
          const double zj = readlane_f64(b, j);
          b = fma(-(tt > j ? Lr[j] : 0.0), zj, b);
        }
      });
      b *= pmf_rcp_f64(dv);
      static_for<1, WVN>([&](auto jr_) {
        constexpr int j = WVN - 1 - decltype(jr_)::value;
        if (j >= shift) {
          const double tot = wave_sum_f64(tt > j ? Lr[j] * b : 0.0);
          if (tt == j) b -= tot;
        }
      });
      if (on_t) vecC[t - shift] = b;
      __syncthreads();
      // ---- z = M[:, S] mu over this lane's 2 variables ----
      double z[2] = {0.0, 0.0};
      wv_rows_dot<false, false>(Msel, KP, t, lst, vecC, ns, z[0], z[1]);
      // ---- candidate solution s and dual w per variable, then block principal pivoting (k_nnqp_quad's rules) ----
      //   complement: P: s = y + z, w = 0;   N: s = 0, w = f - mu(t)
      //   primal:     P: s = mu(t), w = 0;   N: s = 0, w = f - z
      unsigned long long out_m[2], in_m[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bool inP = (pm[s] >> t) & 1ull;
        const bool real = (live[s] >> t) & 1ull;
        const double muv = mypos[s] >= 0 ? vecC[mypos[s]] : 0.0;
        double sv, w;
        if (comp) { sv = inP ? y[s] + z[s] : 0.0; w = (real && !inP) ? f[s] - muv : 0.0; }
        else { sv = inP ? muv : 0.0; w = (real && !inP) ? f[s] - z[s] : 0.0; }
        out_m[s] = __ballot(inP && sv < 0.0);
        in_m[s] = __ballot(real && !inP && w > tol);
        x[s] = inP ? fmax(sv, 0.0) : 0.0;
      }
      __syncthreads();                               // (vecC, vecV, lst are rewritten by the next pass)
      const int ninf = __popcll(out_m[0]) + __popcll(out_m[1]) + __popcll(in_m[0]) + __popcll(in_m[1]);
      ++npass;
      if (ninf == 0 || npass > 6 * k + 16) break;    // KKT holds: x = s on P, zero elsewhere
      bool full = true;
      if (ninf < ninf_best) { ninf_best = ninf; backup = 3; }
      else if (backup > 0) --backup;
      else full = false;
      if (full) {
        pm[0] = (pm[0] & ~out_m[0]) | in_m[0];
        pm[1] = (pm[1] & ~out_m[1]) | in_m[1];
      } else {                                       // Murty: only the infeasible variable with the largest index
        const unsigned long long hi = out_m[1] | in_m[1], lo = out_m[0] | in_m[0];
        if (hi) pm[1] ^= 1ull << (63 - __builtin_clzll(hi));
        else pm[0] ^= 1ull << (63 - __builtin_clzll(lo));
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int v = t + 64 * s;
      if (v < k) X[(int64_t)v * x_sk + prob * x_sp] = (float)(((pm[s] >> t) & 1ull) ? x[s] : 0.0);
    }
  }
}

static inline int launch_nnqp_wave(hipStream_t s, int KP, int k, const double* Horig, const double* Hd, const double* Bd, const float* F,
                                   int64_t f_sk, int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm) {
  if (k <= 64 || k > 128) return PMF_EINVAL;
  int64_t blocks = nprob;
  if (blocks > 256 * 8) blocks = 256 * 8;              // two waves per SIMD (registers)
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_nnqp_wave, dim3((unsigned)blocks), dim3(64), 0, s, Horig, Hd, Bd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm);
  return PMF_OK;
}
