// pmf_nnls_quad.h -- the NMFALS sub-problems (pymf/nmfals.py:70-97) with SIXTEEN LANES PER PROBLEM.
//
//     minimise 1/2 x' HA x - f' x   subject to x >= 0      (HA = H H^T or W^T W, k <= 64 variables)
//
// HA is positive definite here, so the minimiser is unique and characterised by its KKT conditions
// (x >= 0, w = f - HA x <= 0, x w = 0): any exact active-set method ends at the point k_nnqp (pmf_nnls.h) and the
// float64 oracle end at.  k_nnqp walks there one variable at a time and keeps the explicit inverse of HA[P,P] in
// the registers of a whole wave (lane = variable), paying two v_readlane per broadcast operand: 16.6 k VALU
// instructions per problem, two thirds of them broadcasts, the warm start alone 38 rank-one borders.  Here:
//
//  * HA is SHARED by all problems of a half step, so B = inv(HA) is formed ONCE (k_inverse_spd_mfma) and a
//    problem only factorises the smaller of the two complementary blocks -- with P the passive set and
//    N the rest, |P| + |N| = k:
//       |P| <= |N|  (primal):      HA[P,P] x_P = f_P,                     w_N = f_N - HA[N,P] x_P
//       |N| <  |P|  (complement):  x = B (f_P (+) mu),  B[N,N] mu = -(B f_P)_N,   w_N = f_N - mu
//    (the second is the first written for the dual problem: x_N = 0 fixes mu, and HA x = f_P (+) mu makes the
//    multipliers of the zero variables come out of the same solve).  Either way the system has at most 32
//    unknowns, whatever the support of the warm start, and NO border-by-border build-up: a warm start is one
//    factorisation.
//  * that system lives in the registers of a ROW of 16 lanes (4 problems per wave): row i of its LDL^T factor
//    in lane i % 16 (48 doubles per lane), every index a constant of the program text, so the factorisation is
//    straight-line code whose only cross-lane traffic is ONE v_mov_b64_dpp row_newbcast per broadcast operand
//    (register to register: no SGPR round trip, no LDS).
//  * products with HA / B read the shared matrices from LDS (rows padded to 66 doubles and permuted so that a
//    lane's 4 entries of a row are contiguous); operands that are problem-specific come from a small per-problem
//    LDS vector.
//  * the passive set moves by BLOCK principal pivoting (Kim & Park's rule with Murty's single-variable exchange as
//    the safeguard): one solve on the current P gives s_P and w_N; every passive variable with s < 0 leaves and
//    every zero variable with w > tol enters AT ONCE -- from the warm start of an ALS iteration that is two or
//    three solves per problem instead of one per entering / leaving variable.
// Every pass of the wave's loop is ONE solve for each of its 4 problems followed by the exchange; a problem that
// has finished idles until the wave's 4 are done.
//
// TWO FRAMES (template QN, NW): the kernel is a row of latency chains, so its throughput is its waves per SIMD
// (256 / 384 / 512 workgroups of the 32-slot form: 0.84 / 0.62 / 0.48 ms at cfg3).  <32, 4>: the 32-slot frame above, 234
// registers and its own 68 KiB image of HA and B per four-wave workgroup -- two waves per SIMD twice over.  <16, 12>: a
// 16-slot frame (16 doubles per lane, 168 registers) and twelve waves around ONE image -- three waves per SIMD; once the
// active sets have settled a problem factorises about 8 unknowns.  A problem that outgrows 16 puts itself on a list
// and the <32, 4> launch behind solves the list (QuadCtl); bit-identical results (same lanes, same order).
//
// Preconditions (the host checks them, k_nnqp serves the rest): k <= 64, and *warm_flag != 0, i.e.
// k_inverse_spd_mfma's pivots (or k_spd_unique) found HA positive definite and well conditioned -- then B exists and every principal block of
// HA and of B is positive definite too.
#pragma once
#include <type_traits>
#include "pmf_dev.h"
#include "pmf_nnls.h"
#include "pmf_nnls_api.h"   // QuadCtl

constexpr int QLD = 66;            // LDS row stride of the shared matrices in doubles (528 B: rows start on different banks)
constexpr int QNS = 32;            // largest system a problem factorises
constexpr int QPW = 4;             // problems per wave
template <int QN = QNS, int NW = 4>
constexpr size_t nnqp_quad_smem_bytes() {
  return (size_t)2 * 64 * QLD * sizeof(double)                    // HA, B
         + (size_t)NW * QPW * (64 + QN) * sizeof(double)          // per wave and problem: a 64-vector and a QN-vector
         + (size_t)NW * QPW * 128;                                // ... and the P and N lists (bytes)
}

template <int R>
__device__ __forceinline__ double row_bcast(double v) {          // lane R of this 16-lane row, in all 16 lanes
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + R, 0xF, 0xF, false);
}
// reductions over the 16 lanes of a row, result in all of them (DPP butterfly: quad_perm, row_half_mirror, row_mirror)
__device__ __forceinline__ double row_sum(double v) {
  v += dpp_mov_f64<0xB1>(v);
  v += dpp_mov_f64<0x4E>(v);
  v += dpp_mov_f64<0x141>(v);
  v += dpp_mov_f64<0x140>(v);
  return v;
}
__device__ __forceinline__ double row_min(double v) {
  v = fmin(v, dpp_mov_f64<0xB1>(v));
  v = fmin(v, dpp_mov_f64<0x4E>(v));
  v = fmin(v, dpp_mov_f64<0x141>(v));
  return fmin(v, dpp_mov_f64<0x140>(v));
}
__device__ __forceinline__ double row_max(double v) {
  v = fmax(v, dpp_mov_f64<0xB1>(v));
  v = fmax(v, dpp_mov_f64<0x4E>(v));
  v = fmax(v, dpp_mov_f64<0x141>(v));
  return fmax(v, dpp_mov_f64<0x140>(v));
}
template <int CTRL>
__device__ __forceinline__ int dpp_mov_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int row_or(int v) {
  v |= dpp_mov_i32<0xB1>(v);
  v |= dpp_mov_i32<0x4E>(v);
  v |= dpp_mov_i32<0x141>(v);
  v |= dpp_mov_i32<0x140>(v);
  return v;
}
__device__ __forceinline__ int row_min_i(int v) {
  v = min(v, dpp_mov_i32<0xB1>(v));
  v = min(v, dpp_mov_i32<0x4E>(v));
  v = min(v, dpp_mov_i32<0x141>(v));
  return min(v, dpp_mov_i32<0x140>(v));
}

// row block m of a lane (rows 16 m + r, r = lane % 16) holds columns 0 .. 16 m + 15: 16 + 32 = 48 doubles
__host__ __device__ constexpr int qoff(int m) { return 16 * m; }

// for (I = A; I < B; ++I) f(integral_constant<I>) -- the factor's indices must be constants of the program
// text (the array lives in registers), and #pragma unroll gives up on the nested triangular loops
template <int A, int B, typename Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
  if constexpr (A < B) {
    fn(std::integral_constant<int, A>{});
    static_for<A + 1, B>(fn);
  }
}

// (Hp = HA with every DEAD variable -- zero row and column: a basis that has died out -- replaced by the identity, so that
// B = inv(Hp) exists, is formed by k_inverse_spd_mfma itself since round 4: pmf_inv.h, `Gpatched`; such variables never
// become passive -- k_nnqp rejects them when it borders.)

#ifdef PMF_QUAD_COUNT
#define PMF_QSTAMP(idx) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) atomicAdd(&g_quad_t[idx], t_ - tq_); tq_ = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ unsigned long long g_quad_t[8];   // ticks per section: lists, y product, gather, LDL^T, solves, z product, decision
#else
#define PMF_QSTAMP(idx) do { } while (0)
#endif
#ifdef PMF_QUAD_COUNT   // diagnostic build only: wave tasks, passes, sum of the largest system per pass, of the longest product, histogram of sizes / 4
__device__ unsigned long long g_quad_cnt[16];
#endif

// QN: the frame -- the largest system a problem factorises here (32: every problem, k / 2 <= 32; 16: the form for settled
// active sets, 12 waves per workgroup around ONE LDS image of HA and B and a third less registers: three waves per SIMD
// instead of two; a problem that needs more puts itself on a list and is left, untouched, to a QN = 32 launch behind).
// COUNT: the instantiation behind pmf_set_option("nnqp_count", 1) -- it keeps QuadCtl::stats (a counting pass costs 8 % of
// the kernel's time: the production loop carries no counters).  (y0 = B f of all problems as ONE float64-MFMA product ahead
// of the kernel -- k_nnqp_y0, as k_nnqp_wave has it -- was measured and dropped: the product and the 134 MB it writes and
// this kernel reads back cost more than the loop over B's rows below, 0.493 -> 0.526 ms per W half step at cfg3,
// profiles/r04_experiments.md.)
template <int QN, int NW, bool COUNT>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void k_nnqp_quad(const double* __restrict__ Horig, const double* __restrict__ Hd,
                                                      const double* __restrict__ Bd,
                                                      int KP, int k, const float* __restrict__ F, int64_t f_sk,
                                                      int64_t f_sp, float* __restrict__ X, int64_t x_sk, int64_t x_sp,
                                                      int64_t nprob, const int* __restrict__ warm_flag, const QuadCtl ctl) {
  int* const nbig = ctl.nbig;
  const int* const nbig_prev = ctl.nbig_prev;
  constexpr int MB = QN / 16;                        // 16-row blocks of the frame
  // the previous half step of this kind met more than a tenth of its problems beyond a 16-slot frame (the first iterations
  // from a random start): the 16-slot launch hands everything on at once, the 32-slot launch counts for the next decision
  const bool all_big = nbig_prev != nullptr && 10 * (int64_t)(*nbig_prev) > nprob;
  if (QN < 32 && blockIdx.x == 0 && threadIdx.x == 0) {               // (the first launch of a call: the next call's counters)
    if (ctl.dcount_next) *ctl.dcount_next = 0;
    if (ctl.nbig_next) *ctl.nbig_next = 0;
  }
  if (*warm_flag == 0) return;                       // HA not safely positive definite: k_nnqp takes the half step
  if (QN < 32 && all_big) return;                    // everything goes to the 32-slot launch behind
  const bool listed = QN == 32 && ctl.dlist != nullptr && !all_big;   // this launch solves the problems of the list only
  int64_t ntot = nprob;
  if (listed) {
    ntot = *ctl.dcount;
    if (ntot == 0) return;                           // nothing was left over: not even the LDS images are staged
  }
  if ((int64_t)blockIdx.x * NW * QPW >= ntot) return;   // (a short list: the workgroups beyond it leave before staging, too)
  extern __shared__ __attribute__((aligned(16))) double qsm[];
  // [2][64][QLD]: 0 = HA, 1 = B.  Entry (c, t), t = 16 s + r, at c * QLD + (s / 2) * 32 + 2 r + s % 2: lane r's four
  // entries of a row are two 16-byte pieces, and the 16 lanes of a problem read 256 contiguous bytes per piece --
  // all 64 banks once (4 entries side by side per lane made lanes r and r + 8 collide on every read)
  double* sM = qsm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, r = lane & 15;
  double* vecV = qsm + 2 * 64 * QLD + (size_t)(wv * QPW + q) * (64 + QN);    // this problem's 64-vector
  double* vecC = vecV + 64;                                                   // ... and 32-vector
  unsigned char* lst = reinterpret_cast<unsigned char*>(qsm + 2 * 64 * QLD + NW * QPW * (64 + QN)) +
                       (size_t)(wv * QPW + q) * 128;                          // [2][64]: P list, N list
  for (int e = tid; e < 2 * 64 * 64; e += 64 * NW) {
    const int which = e >> 12, c = (e >> 6) & 63, t = e & 63;
    const double* src = which ? Bd : Hd;
    const double v = (c < k && t < k) ? src[(int64_t)c * KP + t] : (c == t ? 1.0 : 0.0);
    sM[which * (64 * QLD) + c * QLD + ((t >> 5) & 1) * 32 + (t & 15) * 2 + ((t >> 4) & 1)] = v;
  }
  __syncthreads();
  double hmax = 0.0;
  for (int t = 0; t < k; ++t) hmax = fmax(hmax, Horig[(int64_t)t * KP + t]);
  const double tol = 2.220446049250313e-15 * (double)k * hmax;       // as k_nnqp and the oracle
  unsigned long long kmask = 0ull;                   // the live variables: real (< k) and not dead
  for (int t = 0; t < k; ++t)
    if (Horig[(int64_t)t * KP + t] > 1e-12 * hmax) kmask |= 1ull << t;
  const unsigned long long live = kmask;
  const int klive = __popcll(kmask);

  const int64_t nwaves = (int64_t)gridDim.x * NW;
  unsigned st_tasks = 0, st_passes = 0, st_ns = 0, st_solved = 0;
  for (int64_t base = ((int64_t)blockIdx.x * NW + wv) * QPW; base < ntot; base += nwaves * QPW) {
    const int64_t slot = base + q;
    const bool valid = slot < ntot;
    const int64_t prob = listed ? (int64_t)ctl.dlist[valid ? slot : 0] : slot;
    bool deferred = false;                           // QN = 16: this problem's system outgrew the frame
    bool ever_big = false;
    // ---- this problem's right-hand side and warm start: variable 16 s + r in slot s of lane r ----
    double f[4], x[4];
    unsigned long long pm = 0ull;
    {
      int lo = 0, hi = 0;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int v = 16 * s + r;
        const bool act = valid && v < k;
        f[s] = act ? (double)F[(int64_t)v * f_sk + prob * f_sp] : 0.0;
        const float x0 = act ? X[(int64_t)v * x_sk + prob * x_sp] : 0.f;
        x[s] = x0 > 0.f ? (double)x0 : 0.0;
        if (x0 > 0.f) { if (v < 32) lo |= 1 << v; else hi |= 1 << (v - 32); }
      }
      lo = row_or(lo); hi = row_or(hi);
      pm = (((unsigned long long)(unsigned)hi << 32) | (unsigned)lo) & kmask;
    }
    bool done = !valid;
    int npass = 0, ninf_best = k + 1, backup = 3;
    // y0 = B (f on the live variables), ONCE per problem: the complement form needs y = B (f on P) in every pass, and
    // P is the large set there (at cfg3 about 56 of 64 variables, 1.9 passes per wave task) -- y = y0 - B (f on N)
    // costs |N| terms per pass instead of |P|.
    double y0[4] = {0.0, 0.0, 0.0, 0.0};
    {
#pragma unroll
      for (int s = 0; s < 4; ++s) vecV[16 * s + r] = f[s];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int c0 = 0; c0 < k; c0 += 4) {
        double fc[4];
        const double* row[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = c0 + e;
          const bool lv = c < k && ((kmask >> c) & 1ull);
          fc[e] = lv ? vecV[c] : 0.0;
          row[e] = sM + 64 * QLD + (c < 64 ? c : 0) * QLD + r * 2;    // B, row c, this lane's two 16-byte pieces
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          y0[0] = fma(row[e][0], fc[e], y0[0]); y0[1] = fma(row[e][1], fc[e], y0[1]);
          y0[2] = fma(row[e][32], fc[e], y0[2]); y0[3] = fma(row[e][33], fc[e], y0[3]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }

    for (int pass = 0; pass < 8 * 64 + 16; ++pass) {
      if (__ballot(!done) == 0ull) break;
#ifdef PMF_QUAD_COUNT
      unsigned long long tq_ = __builtin_amdgcn_s_memtime();
#endif
      // ---- lists of P and N (ascending), the form of the solve ----
      // The system's unknowns sit at positions shift .. 31 of a 32-slot frame (shift = 32 - ns, identity in front):
      // the factorisation can then START at the first position any of the wave's problems uses.
      if (!done) {                                   // (for the next half step's choice of the form: did this problem EVER go beyond 16?)
        const int np0 = __popcll(pm);
        if (min(np0, klive - np0) > 16) ever_big = true;
      }
      if (QN < 32 && !done) {                        // does the smaller of |P|, |N| still fit the frame?
        const int np0 = __popcll(pm);
        if (min(np0, klive - np0) > QN) { deferred = true; done = true; pm = 0ull; }   // (pm = 0: an empty system for the rest of the wave's passes)
      }
      const int np_ = __popcll(pm), nn = klive - np_;
      const bool comp = nn < np_;                    // complement form: factorise B[N,N]
      const int ns = comp ? nn : np_;                // <= min(k / 2, QN)
      const int shift = QN - ns;
      int nsmax = ns, ntmax = comp ? np_ : 0;
#pragma unroll
      for (int o = 32; o >= 16; o >>= 1) {
        nsmax = max(nsmax, __shfl_xor(nsmax, o, 64));
        ntmax = max(ntmax, __shfl_xor(ntmax, o, 64));
      }
      const int jstart = __builtin_amdgcn_readfirstlane(QN - nsmax);
#ifdef PMF_QUAD_COUNT
      if (lane == 0) {
        if (pass == 0) atomicAdd(&g_quad_cnt[0], 1ull);
        atomicAdd(&g_quad_cnt[1], 1ull);
        atomicAdd(&g_quad_cnt[2], (unsigned long long)__builtin_amdgcn_readfirstlane(nsmax));
        atomicAdd(&g_quad_cnt[3], (unsigned long long)__builtin_amdgcn_readfirstlane(ntmax));
        atomicAdd(&g_quad_cnt[4 + min(__builtin_amdgcn_readfirstlane(nsmax) / 4, 8)], 1ull);
      }
#endif
      ntmax = __builtin_amdgcn_readfirstlane(ntmax);
      (void)ntmax;                                   // (diagnostic builds count it: the product over P that is no longer formed)
      if constexpr (COUNT) {
        ++st_passes;
        st_ns += (unsigned)(QN - jstart);
      }
      {
        const int offP = comp ? 0 : shift, offN = comp ? shift : 0;   // the S list right-aligned at byte 32, the other at 0
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int v = 16 * s + r;
          const unsigned long long below = (1ull << v) - 1ull;
          if ((kmask >> v) & 1ull) {
            if ((pm >> v) & 1ull) lst[offP + __popcll(pm & below)] = (unsigned char)v;
            else lst[64 + offN + __popcll(~pm & kmask & below)] = (unsigned char)v;
          }
        }
      }
      // f into the problem's LDS vector (the complement form reads f on P from it, the primal form its right-hand side)
#pragma unroll
      for (int s = 0; s < 4; ++s) vecV[16 * s + r] = f[s];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned char* lS = lst + (comp ? 64 : 0);      // entry of position p at lS[p], p >= shift
      const double* Msel = sM + (comp ? 64 * QLD : 0);
      unsigned pcl[QN / 4];                               // permuted column offsets of the S frame, packed bytes
#pragma unroll
      for (int d = 0; d < QN / 4; ++d) {
        const unsigned sld = reinterpret_cast<const unsigned*>(lS)[d];
        pcl[d] = (sld & 0x20202020u) | ((sld & 0x0f0f0f0fu) << 1) | ((sld >> 4) & 0x01010101u);   // the offset above, per byte
      }

      PMF_QSTAMP(0);
      // ---- complement form, round 4: with nu = mu - f_N the system reads B[N,N] nu = -(y0)_N (y0 = B f, formed once per
      // problem), the solution x = y0 + B[:,N] nu and the multipliers w_N = f_N - mu = -nu: ONE product with the columns of
      // N per pass (below, behind the solve) instead of two -- rounds 2-3 first formed y = B (f on P) = y0 - B[:,N] f_N, a
      // product of its own and one of the seven latency chains of a pass (12 k of its 51 k ticks).  Primal: b = f[P]. ----
      __builtin_amdgcn_wave_barrier();
      if (comp) {
#pragma unroll
        for (int s = 0; s < 4; ++s) vecV[16 * s + r] = y0[s];
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

      PMF_QSTAMP(1);
      // ---- gather M[S,S] into positions shift .. 31 (identity in front) and the right-hand side ----
      double Lr[8 * MB * (MB + 1)], dv[MB], bv[MB];
      static_for<0, MB>([&](auto mc_) {
        constexpr int m = decltype(mc_)::value;
        const int i = 16 * m + r;
        const bool live_i = i >= shift;
        const int si = live_i ? lS[i] : 0;
        const double* rowp = Msel + si * QLD;
        const double vi = live_i ? vecV[si] : 0.0;
        bv[m] = comp ? -vi : vi;
        dv[m] = 1.0;
        static_for<0, 16 * m + 16>([&](auto cc_) {
          constexpr int c = decltype(cc_)::value;
          const int pc = (int)((pcl[c >> 2] >> (8 * (c & 3))) & 0xffu);
          const double val = rowp[c >= shift ? pc : 0];
          Lr[qoff(m) + c] = (live_i && c >= shift) ? val : (c == i ? 1.0 : 0.0);
        });
      });
      PMF_QSTAMP(2);
      // ---- LDL^T, right-looking; row i in lane i % 16, every index a constant of the program text ----
      // The reciprocal of step j + 1's pivot is started as soon as column j + 1 has its step-j update (first in
      // the loop over c): its latency -- v_rcp_f64 and two Newton steps -- runs under the rest of step j's updates.
      double dnext = 1.0, invnext = 1.0;
      static_for<0, QN>([&](auto jc_) {
        constexpr int j = decltype(jc_)::value;
        constexpr int mj = j >> 4, rj = j & 15;
        if (j >= jstart) {                           // (below: identity in every problem of this wave)
          double dj = dnext, inv = invnext;
          if (j == jstart) {
            dj = row_bcast<rj>(Lr[qoff(mj) + j]);
            inv = pmf_rcp_f64(dj);
          }
          if (r == rj) dv[mj] = dj;
          double lij[2];
          static_for<mj, MB>([&](auto mm_) { constexpr int m = decltype(mm_)::value; lij[m] = Lr[qoff(m) + j] * inv; });
          static_for<j + 1, QN>([&](auto cc_) {
            constexpr int c = decltype(cc_)::value;
            constexpr int mc = c >> 4, rc = c & 15;
            const double acj = row_bcast<rc>(Lr[qoff(mc) + j]);
            static_for<mc, MB>([&](auto mm_) {
              constexpr int m = decltype(mm_)::value;
              Lr[qoff(m) + c] = fma(-lij[m], acj, Lr[qoff(m) + c]);
            });
            if (c == j + 1) {                        // the next pivot is final now
              dnext = row_bcast<rc>(Lr[qoff(mc) + c]);
              invnext = pmf_rcp_f64(dnext);
            }
          });
          // (rows <= j of block mj: padding, never read again)
          static_for<mj, MB>([&](auto mm_) { constexpr int m = decltype(mm_)::value; Lr[qoff(m) + j] = lij[m]; });
        }
      });
      PMF_QSTAMP(3);
      // ---- forward: L z = b ----
      static_for<0, QN>([&](auto jc_) {
        constexpr int j = decltype(jc_)::value;
        constexpr int mj = j >> 4, rj = j & 15;
        if (j >= jstart) {
          const double zj = row_bcast<rj>(bv[mj]);
          if (rj < 15) bv[mj] = fma(-(r > rj ? Lr[qoff(mj) + j] : 0.0), zj, bv[mj]);
          static_for<mj + 1, MB>([&](auto mm_) { constexpr int m = decltype(mm_)::value; bv[m] = fma(-Lr[qoff(m) + j], zj, bv[m]); });
        }
      });
      static_for<0, MB>([&](auto mm_) { constexpr int m = decltype(mm_)::value; bv[m] = bv[m] * pmf_rcp_f64(dv[m]); });
      // ---- backward: L^T mu = z ----
      static_for<0, QN>([&](auto jr_) {
        constexpr int j = QN - 1 - decltype(jr_)::value;
        constexpr int mj = j >> 4, rj = j & 15;
        if (j >= jstart) {
          double part = 0.0;
          if (rj < 15) part = (r > rj ? Lr[qoff(mj) + j] : 0.0) * bv[mj];
          static_for<mj + 1, MB>([&](auto mm_) { constexpr int m = decltype(mm_)::value; part = fma(Lr[qoff(m) + j], bv[m], part); });
          const double tot = row_sum(part);
          if (r == rj) bv[mj] -= tot;
        }
      });
      // mu (by position) -> vecC
#pragma unroll
      for (int m = 0; m < MB; ++m) vecC[16 * m + r] = bv[m];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

      PMF_QSTAMP(4);
      // ---- z = M[:, S] mu over this lane's 4 variables ----
      double z[4] = {0.0, 0.0, 0.0, 0.0};
      for (int p0 = jstart & ~3; p0 < QN; p0 += 4) {     // four positions at a time, as above
        const unsigned u = reinterpret_cast<const unsigned*>(lS)[p0 >> 2];
        double mu[4];
        const double* row[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool live = p0 + e >= shift;
          const int c = live ? (int)((u >> (8 * e)) & 0xffu) : 0;
          mu[e] = live ? vecC[p0 + e] : 0.0;
          row[e] = Msel + c * QLD + r * 2;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          z[0] = fma(row[e][0], mu[e], z[0]); z[1] = fma(row[e][1], mu[e], z[1]);
          z[2] = fma(row[e][32], mu[e], z[2]); z[3] = fma(row[e][33], mu[e], z[3]);
        }
      }
      PMF_QSTAMP(5);
      // ---- candidate solution s and dual w per variable ----
      //   complement: P: s = y0 + z, w = 0;  N: s = 0, w = -nu(t)
      //   primal:     P: s = mu(t), w = 0;   N: s = 0, w = f - z
      double sv[4], w[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int v = 16 * s + r;
        const bool inP = (pm >> v) & 1ull;
        const bool real = (kmask >> v) & 1ull;
        const unsigned long long below = (1ull << v) - 1ull;
        const unsigned long long smask = comp ? (~pm & kmask) : pm;
        const bool inS = (smask >> v) & 1ull;
        const double muv = inS ? vecC[(shift + __popcll(smask & below)) & (QN - 1)] : 0.0;
        if (comp) { sv[s] = inP ? y0[s] + z[s] : 0.0; w[s] = (real && !inP) ? -muv : 0.0; }
        else { sv[s] = inP ? muv : 0.0; w[s] = (real && !inP) ? f[s] - z[s] : 0.0; }
      }
      __builtin_amdgcn_wave_barrier();

      // ---- block principal pivoting: all infeasible variables change sides at once ----
      if (!done) {
        int outlo = 0, outhi = 0, inlo = 0, inhi = 0, ninf_l = 0, top = -1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int v = 16 * s + r;
          const bool inP = (pm >> v) & 1ull;
          const bool leave = inP && sv[s] < 0.0;
          const bool enter = ((live >> v) & 1ull) && !inP && w[s] > tol;
          if (leave) { if (v < 32) outlo |= 1 << v; else outhi |= 1 << (v - 32); }
          if (enter) { if (v < 32) inlo |= 1 << v; else inhi |= 1 << (v - 32); }
          if (leave || enter) { ++ninf_l; top = v; }           // (v ascends with s: the lane's largest)
          x[s] = inP ? fmax(sv[s], 0.0) : 0.0;
        }
        outlo = row_or(outlo); outhi = row_or(outhi); inlo = row_or(inlo); inhi = row_or(inhi);
        const unsigned long long out_m = ((unsigned long long)(unsigned)outhi << 32) | (unsigned)outlo;
        const unsigned long long in_m = ((unsigned long long)(unsigned)inhi << 32) | (unsigned)inlo;
        const int ninf = __popcll(out_m) + __popcll(in_m);
        ++npass;
        if (ninf == 0 || npass > 6 * k + 16) {
          done = true;                               // KKT holds: x = s on P, zero elsewhere
        } else {
          bool full = true;
          if (ninf < ninf_best) { ninf_best = ninf; backup = 3; }
          else if (backup > 0) --backup;
          else full = false;
          if (full) {
            pm = (pm & ~out_m) | in_m;
          } else {                                   // Murty: only the infeasible variable with the largest index
            const int vtop = 63 - __builtin_clzll(out_m | in_m);
            pm ^= 1ull << vtop;
          }
        }
        (void)ninf_l; (void)top;
      }
      PMF_QSTAMP(6);
      __builtin_amdgcn_wave_barrier();
    }
    if (valid && !deferred) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int v = 16 * s + r;
        if (v < k) X[(int64_t)v * x_sk + prob * x_sp] = (float)(((pm >> v) & 1ull) ? x[s] : 0.0);
      }
    }
    if constexpr (COUNT) {
      ++st_tasks;
      st_solved += (unsigned)__popcll(__ballot(valid && !deferred && r == 0));
    }
    if (QN < 32 && ctl.dlist != nullptr && deferred && r == 0) ctl.dlist[atomicAdd(ctl.dcount, 1)] = (int)prob;
    // counted by the launch that sees the problem first (the listed ones were counted when they were put on the list)
    if (nbig != nullptr && valid && ever_big && r == 0 && (QN < 32 || ctl.dlist == nullptr || all_big)) atomicAdd(nbig, 1);
  }
  if (COUNT && ctl.stats != nullptr && lane == 0 && st_tasks != 0) {
    unsigned long long* st = ctl.stats + (QN < 32 ? 0 : 4);
    atomicAdd(st + 0, (unsigned long long)st_tasks); atomicAdd(st + 1, (unsigned long long)st_passes);
    atomicAdd(st + 2, (unsigned long long)st_ns); atomicAdd(st + 3, (unsigned long long)st_solved);
  }
}

#ifndef PMF_QUAD_TEMPLATES_ONLY   // (pmf_nnls_tu.hip includes this header for static_for only)
template <int QN, int NW, bool COUNT>
static inline int launch_nnqp_quad_t(hipStream_t s, int KP, int k, const double* Horig, const double* Hd, const double* Bd, const float* F, int64_t f_sk,
                                     int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm, const QuadCtl& ctl) {
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[pmf_current_device()];
  const size_t smem = nnqp_quad_smem_bytes<QN, NW>();
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nnqp_quad<QN, NW, COUNT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return PMF_EHIP;
    attr_done = true;
  }
  int64_t blocks = (nprob + NW * QPW - 1) / (NW * QPW);
  const int64_t cap = NW == 4 ? 512 : 256;             // two workgroups of 4 waves, or one of 12, per CU
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((k_nnqp_quad<QN, NW, COUNT>), dim3((unsigned)blocks), dim3(64 * NW), smem, s, Horig, Hd, Bd, KP, k, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm,
                     ctl);
  return PMF_OK;
}

// ctl == nullptr: every problem on the 32-slot frame (one launch).  Else: the 16-slot frame first (three waves per SIMD),
// then the 32-slot frame for the problems it listed.  count: the counting instantiations (QuadCtl::stats).
static inline int launch_nnqp_quad(hipStream_t s, int KP, int k, const double* Horig, const double* Hd, const double* Bd, const float* F, int64_t f_sk,
                                   int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, const int* warm, const QuadCtl* ctl = nullptr,
                                   bool count = false) {
  if (ctl) {
    const int rc = count ? launch_nnqp_quad_t<16, 12, true>(s, KP, k, Horig, Hd, Bd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, *ctl)
                         : launch_nnqp_quad_t<16, 12, false>(s, KP, k, Horig, Hd, Bd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, *ctl);
    if (rc != PMF_OK) return rc;
    return count ? launch_nnqp_quad_t<32, 4, true>(s, KP, k, Horig, Hd, Bd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, *ctl)
                 : launch_nnqp_quad_t<32, 4, false>(s, KP, k, Horig, Hd, Bd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm, *ctl);
  }
  return launch_nnqp_quad_t<32, 4, false>(s, KP, k, Horig, Hd, Bd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, warm,
                                          QuadCtl{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr});
}
#endif
