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
//     factor in a 64 x 65 LDS image (left-looking, 35 KiB: one wave per SIMD) took 7.5-12 ms per iteration where this
//     one takes 2.1-4.9 (65 536 x 512, k = 128; profiles/r03_experiments.md).  The backward solve runs column by column
//     from a packed copy of L's rows in LDS (16 KiB per wave);
//   * products with HA / B read rows of the shared matrices from L2 (128 KiB each: beyond LDS), coalesced, sixteen rows
//     per round trip; y0 = B f of ALL problems is one float64-MFMA product (k_nnqp_y0 below);
//   * a problem's inputs are requested one problem ahead.
// Preconditions as k_nnqp_quad: *warm_flag != 0 (k_inverse_spd_mfma's pivots found HA positive definite and well
// conditioned); otherwise the kernel returns at once and k_nnqp_big takes the half step.
#pragma once
#include "pmf_dev.h"
#include "pmf_nnls.h"
#include "pmf_nnls_quad.h"   // static_for

#ifndef PMF_WAVE_NWV
#define PMF_WAVE_NWV 1
#endif
constexpr int WVN = 64;            // largest system a problem factorises (k <= 128)

// a0 += sum_p coef(p) M[row(p)][t], a1 += ... M[row(p)][t + 64] over the positions p < ns of the system's list; coef(p) is
// cv[row(p)] (BYVAR) or cv[p].  SIXTEEN rows -- 32 requests -- go out together: the rows come from L2 (the matrices are
// 128 KiB each) and a SIMD holds two such waves, so the requests in flight are most of the latency hiding
// (profiles/r03_experiments.md).
template <bool BYVAR, bool NEG>
__device__ __forceinline__ void wv_rows_dot(const double* __restrict__ M, int KP, int t, const int* lst, const double* cv, int ns,
                                            double& a0, double& a1) {
  constexpr int R = 16;            // rows per round trip (8: 2 x 16 requests in flight, twice the trips; the factor's registers are dead here)
  const char* base = reinterpret_cast<const char*>(M) + (size_t)t * 8;
  for (int p = 0; p < ns; p += R) {
    int c[R];
    double v0[R], v1[R];
#pragma unroll
    for (int e = 0; e < R; ++e) c[e] = p + e < ns ? lst[p + e] : 0;
#pragma unroll
    for (int e = 0; e < R; ++e) {  // (a 32-bit offset from the uniform base: one address register per row)
      const char* r = base + (unsigned)(c[e] * KP) * 8u;
      v0[e] = *reinterpret_cast<const double*>(r);
      v1[e] = *reinterpret_cast<const double*>(r + 512);
    }
#pragma unroll
    for (int e = 0; e < R; ++e) {
      const double mv = p + e < ns ? cv[BYVAR ? c[e] : p + e] : 0.0;
      const double m = NEG ? -mv : mv;
      a0 = fma(v0[e], m, a0); a1 = fma(v1[e], m, a1);
    }
  }
}

// Y0[prob][v] = sum_c B[c][v] f(c, prob): the complement form's y0 = B f for ALL problems of a half step as one product
// on the float64 MFMA (one wave per 16 problems x 16 variables, operands straight from L2 in MFMA operand order, 32
// requests in flight), instead of 128 rows of B per problem inside k_nnqp_wave (128 KiB of L2 reads per problem: 45 % of
// that kernel's time).  Dead variables need no mask: their rows of B are unit vectors.  grid = (KP / 64, ceil(nprob / 16)).
__global__ __launch_bounds__(256) void k_nnqp_y0(const double* __restrict__ Bd, int KP, const float* __restrict__ F, int64_t f_sk,
                                                 int64_t f_sp, int64_t nprob, double* __restrict__ Y0, const int* __restrict__ warm_flag) {
  if (*warm_flag == 0) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int c0 = (blockIdx.x * 4 + wv) * 16;
  const int64_t p0 = (int64_t)blockIdx.y * 16;
  const int64_t pa = p0 + i < nprob ? p0 + i : nprob - 1;                 // (rows beyond the last problem: computed, not stored)
  const float* ap = F + pa * f_sp + (int64_t)g * f_sk;                    // f(4 s + g, p0 + i)
  const double* bp = Bd + (int64_t)g * KP + c0 + i;                       // B[4 s + g][c0 + i]
  f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
  for (int s0 = 0; s0 < KP / 4; s0 += 16) {
    double a[16], b[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { a[u] = (double)ap[(int64_t)4 * (s0 + u) * f_sk]; b[u] = bp[(int64_t)4 * (s0 + u) * KP]; }
#pragma unroll
    for (int u = 0; u < 16; u += 2) {
      acc0 = mfma_f64(a[u], b[u], acc0);
      acc1 = mfma_f64(a[u + 1], b[u + 1], acc1);
    }
  }
  const f64x4 acc = acc0 + acc1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t prob = p0 + g + 4 * r;
    if (prob < nprob) Y0[prob * KP + c0 + i] = acc[r];
  }
}

template <int NWV>   // waves (problems) per workgroup (1: see below)
__global__ __launch_bounds__(64 * NWV, 2) void k_nnqp_wave(const double* __restrict__ Horig, const double* __restrict__ Hd,
                                                  const double* __restrict__ Bd, int KP, int k,
                                                  const float* __restrict__ F, int64_t f_sk, int64_t f_sp,
                                                  float* __restrict__ X, int64_t x_sk, int64_t x_sp, int64_t nprob,
                                                  const int* __restrict__ warm_flag, const double* __restrict__ Y0) {
  if (*warm_flag == 0) return;
  __shared__ double s_vecP[NWV][2 * WVN]; // the pivot column of a factorisation step, double buffered
  __shared__ double s_vecV[NWV][128];     // f, then (complement form) y, by variable
  __shared__ double s_vecC[NWV][WVN];     // mu by position
  __shared__ int s_lst[NWV][WVN];         // the variable at each position of the system
  __shared__ double s_R[NWV][WVN * (WVN - 1) / 2 + WVN];   // L by rows, packed (row i: i entries at i (i - 1) / 2): the backward solve's operand
  const int t = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double* const vecP = s_vecP[wv];
  double* const vecV = s_vecV[wv];
  double* const vecC = s_vecC[wv];
  int* const lst = s_lst[wv];
  double* const R = s_R[wv];
  const int tri = t * (t - 1) / 2;
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

  // The factorisation and the solves are ~90 KiB of straight-line code, more than the instruction cache two CUs share.
  // NWV > 1 makes the waves of a workgroup START every pass together (one barrier), so that a line fetched by the first
  // serves the others -- measured with NWV = 8: 20-40 % SLOWER (waves idle for the slowest problem's passes; instruction fetch is
  // not the limit: waves wait on memory 6 % of their cycles, the VALU is busy 40 % with two waves of dependent float64 chains
  // per SIMD, profiles/r03_experiments.md).  NWV = 1 is the form in use: every wave on its own.
  // A problem's inputs (right-hand side, warm start, its row of Y0: HBM, ~2 us away) are requested one problem AHEAD, under
  // the previous problem's passes.
  float f_nx[2], x_nx[2];
  double y0_nx[2];
  auto request = [&](int64_t pr) {
    const bool ok = pr < nprob;
    const int64_t p = ok ? pr : nprob - 1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int v = t + 64 * s;
      const bool act = ok && v < k;
      f_nx[s] = act ? F[(int64_t)v * f_sk + p * f_sp] : 0.f;
      x_nx[s] = act ? X[(int64_t)v * x_sk + p * x_sp] : 0.f;
      y0_nx[s] = Y0[p * KP + v];
    }
  };
  request((int64_t)blockIdx.x * NWV + wv);
  for (int64_t pbase = (int64_t)blockIdx.x * NWV; pbase < nprob; pbase += (int64_t)gridDim.x * NWV) {
    const bool valid = pbase + wv < nprob;
    const int64_t prob = valid ? pbase + wv : nprob - 1;
    double f[2], x[2];
    unsigned long long pm[2];
    const double y0[2] = {y0_nx[0], y0_nx[1]};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      f[s] = (double)f_nx[s];
      const float x0 = x_nx[s];
      x[s] = x0 > 0.f ? (double)x0 : 0.0;
      pm[s] = __ballot(x0 > 0.f) & live[s];
    }
    request(pbase + (int64_t)gridDim.x * NWV + wv);      // (a problem of this wave's own, later: nobody writes its X before)
    int npass = 0, ninf_best = k + 1, backup = 3;

    bool done = !valid;
    for (int pass = 0; pass < 8 * 128 + 16; ++pass) {
      if (__syncthreads_or(done ? 0 : 1) == 0) break;
      if (done) continue;
#ifdef PMF_QUAD_COUNT
      const int lane = t;
      unsigned long long tq_ = __builtin_amdgcn_s_memtime();
#endif
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
      __builtin_amdgcn_wave_barrier();
      const double* __restrict__ Msel = comp ? Bd : Hd;
#ifdef PMF_QUAD_COUNT
      if (t == 0) { if (pass == 0) atomicAdd(&g_quad_cnt[0], 1ull); atomicAdd(&g_quad_cnt[1], 1ull); atomicAdd(&g_quad_cnt[2], (unsigned long long)ns); atomicAdd(&g_quad_cnt[4 + min(ns / 8, 8)], 1ull); }
#endif
      PMF_QSTAMP(0);
      // ---- complement form (round 4, as k_nnqp_quad): with nu = mu - f_N the system is B[N,N] nu = -(y0)_N, y0 = B f once per
      //      problem (k_nnqp_y0); x = y0 + B[:,N] nu, w_N = -nu: ONE product with the rows of N per pass (behind the solve)
      //      instead of two (rounds 2-3 first formed y = y0 - B (f on N): ns rows of B from L2 per pass) ----
      if (comp) {
        __builtin_amdgcn_wave_barrier();
        vecV[t] = y0[0]; vecV[t + 64] = y0[1];
        __builtin_amdgcn_wave_barrier();
      }
      PMF_QSTAMP(1);
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
      static_for<0, WVN / 8>([&](auto gc_) {           // eight columns at a time: chunks in front of the system are identity, no requests
        constexpr int c8 = 8 * decltype(gc_)::value;
        if (c8 + 7 >= shift) {
          static_for<c8, c8 + 8>([&](auto cc_) {
            constexpr int c = decltype(cc_)::value;
            const int rv = c >= shift ? lst[c - shift] : 0;
            // (a 32-bit offset from the uniform base: one address register per request, not two)
            const double val = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(Msel) + (unsigned)(rv * KP + myvar) * 8u);
            Lr[c] = (on_t && c >= shift) ? val : (c == tt ? 1.0 : 0.0);
          });
        } else {
          static_for<c8, c8 + 8>([&](auto cc_) { constexpr int c = decltype(cc_)::value; Lr[c] = (c == tt ? 1.0 : 0.0); });
        }
      });
      PMF_QSTAMP(2);
      // ---- LDL^T, right-looking: step j scales column j and takes l_ij a_cj off every later column c; a_cj = A[c][j]
      //      (symmetry: lane c's own entry j) reaches all lanes through a 64-double LDS line, ONE write and broadcast reads ----
      //      Column j + 1 is final after its step-j update: it is published (and the next pivot read back) FIRST, so that the
      //      LDS round trip and the reciprocal run under the rest of step j's updates.
      double dv = 1.0, dnext = 1.0;
      static_for<0, WVN>([&](auto jc_) {
        constexpr int j = decltype(jc_)::value;
        if (j >= shift) {
          double* col = vecP + (j & 1) * WVN;
          double dj = dnext;
          if (j == shift) {                            // the first step: nobody has published its column
            col[t] = Lr[j];
            __builtin_amdgcn_wave_barrier();
            dj = col[j];
          }
          const double inv = pmf_rcp_f64(dj);
          const double lij = Lr[j] * inv;
          if (tt == j) dv = dj;
          if (tt > j) R[tri + j] = lij;
          if constexpr (j + 1 < WVN) {
            double* coln = vecP + ((j + 1) & 1) * WVN;
            Lr[j + 1] = fma(-lij, col[j + 1], Lr[j + 1]);
            coln[t] = Lr[j + 1];
            __builtin_amdgcn_wave_barrier();             // (one wave: its LDS operations complete in order)
            dnext = coln[j + 1];
          }
          static_for<j + 2, WVN>([&](auto cc_) {
            constexpr int c = decltype(cc_)::value;
            Lr[c] = fma(-lij, col[c], Lr[c]);
          });
          Lr[j] = lij;
        }
      });
      PMF_QSTAMP(3);
      // ---- L z = b, z / d, L^T mu = z ----
      static_for<0, WVN - 1>([&](auto jc_) {
        constexpr int j = decltype(jc_)::value;
        if (j >= shift) {
          const double zj = readlane_f64(b, j);
          b = fma(-(tt > j ? Lr[j] : 0.0), zj, b);
        }
      });
      b *= pmf_rcp_f64(dv);
      // (column-oriented, from the packed copy of L's rows in LDS: mu_i is final when its turn comes, row i's entries are
      //  contiguous over the lanes.  The row-oriented form on the registers -- a 64-lane sum per unknown -- was 30
      //  instructions per step instead of 6.)
      __builtin_amdgcn_wave_barrier();
      static_for<0, WVN - 1>([&](auto ir_) {
        constexpr int i = WVN - 1 - decltype(ir_)::value;      // 63 .. 1
        if (i > shift) {
          const double mi = readlane_f64(b, i);
          const double l = R[i * (i - 1) / 2 + t];
          if (tt < i && tt >= shift) b = fma(-l, mi, b);
        }
      });
      if (on_t) vecC[t - shift] = b;
      __builtin_amdgcn_wave_barrier();
      PMF_QSTAMP(4);
      // ---- z = M[:, S] mu over this lane's 2 variables ----
      double z[2] = {0.0, 0.0};
      wv_rows_dot<false, false>(Msel, KP, t, lst, vecC, ns, z[0], z[1]);
      PMF_QSTAMP(5);
      // ---- candidate solution s and dual w per variable, then block principal pivoting (k_nnqp_quad's rules) ----
      //   complement: P: s = y0 + z, w = 0;  N: s = 0, w = -nu(t)
      //   primal:     P: s = mu(t), w = 0;   N: s = 0, w = f - z
      unsigned long long out_m[2], in_m[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bool inP = (pm[s] >> t) & 1ull;
        const bool real = (live[s] >> t) & 1ull;
        const double muv = mypos[s] >= 0 ? vecC[mypos[s]] : 0.0;
        double sv, w;
        if (comp) { sv = inP ? y0[s] + z[s] : 0.0; w = (real && !inP) ? -muv : 0.0; }
        else { sv = inP ? muv : 0.0; w = (real && !inP) ? f[s] - z[s] : 0.0; }
        out_m[s] = __ballot(inP && sv < 0.0);
        in_m[s] = __ballot(real && !inP && w > tol);
        x[s] = inP ? fmax(sv, 0.0) : 0.0;
      }
      __builtin_amdgcn_wave_barrier();                               // (vecC, vecV, lst are rewritten by the next pass)
      PMF_QSTAMP(6);
      const int ninf = __popcll(out_m[0]) + __popcll(out_m[1]) + __popcll(in_m[0]) + __popcll(in_m[1]);
      ++npass;
      if (ninf == 0 || npass > 6 * k + 16) { done = true; continue; }   // KKT holds: x = s on P, zero elsewhere
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
      if (valid && v < k) X[(int64_t)v * x_sk + prob * x_sp] = (float)(((pm[s] >> t) & 1ull) ? x[s] : 0.0);
    }
  }
}

static inline int launch_nnqp_wave(hipStream_t s, int KP, int k, const double* Horig, const double* Hd, const double* Bd, const float* F,
                                   int64_t f_sk, int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm,
                                   double* Y0 /* [nprob][KP] */) {
  if (k <= 64 || k > 128 || KP != 128 || !Y0) return PMF_EINVAL;
  hipLaunchKernelGGL(k_nnqp_y0, dim3(KP / 64, (unsigned)((nprob + 15) / 16)), dim3(256), 0, s, Bd, KP, F, f_sk, f_sp, nprob, Y0, warm);
  constexpr int NWV = PMF_WAVE_NWV;
  int64_t blocks = (nprob + NWV - 1) / NWV;
  if (blocks > 256 * 8 / NWV) blocks = 256 * 8 / NWV;  // two waves per SIMD (registers)
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((k_nnqp_wave<NWV>), dim3((unsigned)blocks), dim3(64 * NWV), 0, s, Horig, Hd, Bd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, Y0);
  return PMF_OK;
}
