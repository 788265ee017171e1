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
//   * the system has at most k / 2 <= 64 unknowns whatever the support: its LDL^T lives in a 64 x 65 LDS image of the
//     wave's own (left-looking, lane = row: a column is one register accumulation over two LDS reads per term, the
//     pivot row U = L D kept in the upper triangle so that its entry is ONE broadcast read);
//   * products with HA / B read rows of the shared matrices from L2 (128 KiB each: beyond LDS), coalesced.
// Preconditions as k_nnqp_quad: *warm_flag != 0 (k_inverse_spd_mfma's pivots found HA positive definite and well
// conditioned); otherwise the kernel returns at once and k_nnqp_big takes the half step.
#pragma once
#include "pmf_dev.h"
#include "pmf_nnls.h"

constexpr int WVN = 64;            // largest system a problem factorises (k <= 128)
constexpr int WVLD = 65;           // LDS row stride of the factor in doubles

__global__ __launch_bounds__(64) void k_nnqp_wave(const double* __restrict__ Horig, const double* __restrict__ Hd,
                                                  const double* __restrict__ Bd, int KP, int k,
                                                  const float* __restrict__ F, int64_t f_sk, int64_t f_sp,
                                                  float* __restrict__ X, int64_t x_sk, int64_t x_sp, int64_t nprob,
                                                  const int* __restrict__ warm_flag) {
  if (*warm_flag == 0) return;
  __shared__ double S[WVN * WVLD];
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
          double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
          for (int c = 0; c < k; c += 2) {
            const bool l0 = (live[c >> 6] >> (c & 63)) & 1ull, l1 = c + 1 < k && ((live[(c + 1) >> 6] >> ((c + 1) & 63)) & 1ull);
            const double f0 = l0 ? vecV[c] : 0.0, f1 = l1 ? vecV[c + 1] : 0.0;
            const double* r0 = Bd + (int64_t)c * KP + t;
            const double* r1 = Bd + (int64_t)(c + 1 < k ? c + 1 : c) * KP + t;
            a0 = fma(r0[0], f0, a0); a1 = fma(r0[64], f0, a1);
            b0 = fma(r1[0], f1, b0); b1 = fma(r1[64], f1, b1);
          }
          y0[0] = a0 + b0; y0[1] = a1 + b1;
          have_y0 = true;
        }
        double a0 = y0[0], a1 = y0[1], b0 = 0.0, b1 = 0.0;
        int p = 0;
        for (; p + 1 < ns; p += 2) {
          const int c0 = lst[p], c1 = lst[p + 1];
          const double f0 = vecV[c0], f1 = vecV[c1];
          const double* r0 = Bd + (int64_t)c0 * KP + t;
          const double* r1 = Bd + (int64_t)c1 * KP + t;
          a0 = fma(-r0[0], f0, a0); a1 = fma(-r0[64], f0, a1);
          b0 = fma(-r1[0], f1, b0); b1 = fma(-r1[64], f1, b1);
        }
        if (p < ns) {
          const int c0 = lst[p];
          const double f0 = vecV[c0];
          const double* r0 = Bd + (int64_t)c0 * KP + t;
          a0 = fma(-r0[0], f0, a0); a1 = fma(-r0[64], f0, a1);
        }
        y[0] = a0 + b0; y[1] = a1 + b1;
        __syncthreads();
        vecV[t] = y[0]; vecV[t + 64] = y[1];
        __syncthreads();
      }
      // ---- the system: M[S,S] into the LDS image (row = position), the right-hand side by position in lane = position ----
      const int myvar = t < ns ? lst[t] : 0;
      double b = 0.0;
      if (t < ns) b = comp ? -vecV[myvar] : vecV[myvar];
      {
        int i = 0;
        for (; i + 3 < ns; i += 4) {
          const int r0 = lst[i], r1 = lst[i + 1], r2 = lst[i + 2], r3 = lst[i + 3];
          const double v0 = Msel[(int64_t)r0 * KP + myvar], v1 = Msel[(int64_t)r1 * KP + myvar];
          const double v2 = Msel[(int64_t)r2 * KP + myvar], v3 = Msel[(int64_t)r3 * KP + myvar];
          S[(i + 0) * WVLD + t] = v0; S[(i + 1) * WVLD + t] = v1; S[(i + 2) * WVLD + t] = v2; S[(i + 3) * WVLD + t] = v3;
        }
        for (; i < ns; ++i) S[i * WVLD + t] = Msel[(int64_t)lst[i] * KP + myvar];
      }
      __syncthreads();
      // ---- LDL^T, left-looking: column j of lane t >= j is a_tj - sum_p L[t][p] U[j][p], U[j][p] = d_p L[j][p] at S[p][j] ----
      double dv = 1.0;
      for (int j = 0; j < ns; ++j) {
        const double* Lrow = S + t * WVLD;
        const double* Ucol = S + j;
        double a0 = Lrow[j], a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int p = 0;
        for (; p + 3 < j; p += 4) {
          a0 = fma(-Lrow[p], Ucol[p * WVLD], a0);
          a1 = fma(-Lrow[p + 1], Ucol[(p + 1) * WVLD], a1);
          a2 = fma(-Lrow[p + 2], Ucol[(p + 2) * WVLD], a2);
          a3 = fma(-Lrow[p + 3], Ucol[(p + 3) * WVLD], a3);
        }
        for (; p < j; ++p) a0 = fma(-Lrow[p], Ucol[p * WVLD], a0);
        const double a = (a0 + a1) + (a2 + a3);
        const double dj = readlane_f64(a, j);
        const double inv = pmf_rcp_f64(dj);
        if (t == j) dv = dj;
        if (t > j && t < ns) {                       // (row j beyond the diagonal and column j below it: read by no lane in this step)
          S[j * WVLD + t] = a;                       // U[t][j]
          S[t * WVLD + j] = a * inv;                 // L[t][j]
        }
        __syncthreads();
      }
      // ---- L z = b, z / d, L^T mu = z ----
      for (int j = 0; j < ns; ++j) {
        const double zj = readlane_f64(b, j);
        if (t > j && t < ns) b = fma(-S[t * WVLD + j], zj, b);
      }
      b *= pmf_rcp_f64(dv);
      for (int j = ns - 1; j >= 0; --j) {
        const double mj = readlane_f64(b, j);
        if (t < j) b = fma(-S[j * WVLD + t], mj, b);
      }
      if (t < ns) vecC[t] = b;
      __syncthreads();
      // ---- z = M[:, S] mu over this lane's 2 variables ----
      double z[2];
      {
        double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
        int p = 0;
        for (; p + 1 < ns; p += 2) {
          const int c0 = lst[p], c1 = lst[p + 1];
          const double m0 = vecC[p], m1 = vecC[p + 1];
          const double* r0 = Msel + (int64_t)c0 * KP + t;
          const double* r1 = Msel + (int64_t)c1 * KP + t;
          a0 = fma(r0[0], m0, a0); a1 = fma(r0[64], m0, a1);
          b0 = fma(r1[0], m1, b0); b1 = fma(r1[64], m1, b1);
        }
        if (p < ns) {
          const int c0 = lst[p];
          const double m0 = vecC[p];
          const double* r0 = Msel + (int64_t)c0 * KP + t;
          a0 = fma(r0[0], m0, a0); a1 = fma(r0[64], m0, a1);
        }
        z[0] = a0 + b0; z[1] = a1 + b1;
      }
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
  if (blocks > 256 * 4) blocks = 256 * 4;              // four 35 KiB images per CU
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_nnqp_wave, dim3((unsigned)blocks), dim3(64), 0, s, Horig, Hd, Bd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm);
  return PMF_OK;
}
