// pmf_nndsvd.h -- NNDSVD initialisation of W and H (pymf/nndsvd.py:79-108), SURVEY 8(f) row 4.
//
// The reference takes the SVD of the data through the eigen-decomposition of the Gram matrix
// (pymf/svd.py:125-148, `_left_svd`: AA = data^T data, eigh, U = data V S^-1) and then, for every
// basis i >= 1, a SECOND full SVD of the positive part of the rank-one matrix s_i u_i v_i^T
// (nndsvd.py:92-106).  That positive part is u+ v+^T + u- v-^T with disjoint supports, so its
// leading singular triple is known in closed form (Boutsidis & Gallopoulos 2008, the paper the
// reference cites): sigma = s_i max(|u+||v+|, |u-||v-|) with the matching normalised parts.
// The device path therefore is
//   1. A = V^T V            k_colgemm on column blocks of V (fp32 MFMA, float64 slab sums)
//   2. A = Q diag(l) Q^T    k_jacobi_eigh: parallel-order two-sided Jacobi in float64
//   3. top-k (l > 1e-8, svd.py:130-131), s = sqrt(l), basis rows v_i / s_i
//   4. U = V (v_i / s_i)    k_rowgemm<EPI_STORE> (fp32 MFMA)
//   5. |u+|^2, |u-|^2 per column (float64), closed form -> H rows and the per-column W scaling.
#pragma once
#include <hip/hip_cooperative_groups.h>
#include "pmf_dev.h"

// Ad[(c0 + r) * ld + c] = sum over slabs of slab[s][r][c]  (r < KPb, c < np), float64, fixed order.
__global__ __launch_bounds__(256) void k_gram_reduce(const float* __restrict__ slab, int nslabs, int KPb,
                                                     int np, int c0, double* __restrict__ Ad, int ld) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= KPb * np) return;
  const int r = idx / np, c = idx % np;
  const int64_t ldp = (int64_t)np + KPb;
  double s = 0.0;
  for (int sl = 0; sl < nslabs; ++sl) s += (double)slab[((int64_t)sl * KPb + r) * ldp + c];
  Ad[(int64_t)(c0 + r) * ld + c] = s;
}

// A[r][c] = A[c][r] for r > c (n x n, leading dimension n): the Gram passes form the upper block triangle only
__global__ __launch_bounds__(256) void k_mirror_upper_f64(double* __restrict__ A, int n) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)n * n) return;
  const int r = (int)(q / n), c = (int)(q % n);
  if (r > c) A[q] = A[(int64_t)c * n + r];
}

// Symmetric eigen-decomposition of the nj x nj (nj even, <= PMF_NNDSVD_MAX_N) matrix A (leading dimension ld):
// cyclic two-sided Jacobi in the round-robin parallel order, float64.  Every step rotates nj/2
// disjoint (p, q) pairs at once: A <- J^T A J decomposes into independent 2 x 2 blocks
// A[{p,q}][{p',q'}] <- J_pq^T (.) J_p'q', one block per thread (the upper triangle of blocks is
// computed, the mirror image written, so A stays exactly symmetric), and QT <- J^T QT row pairs.
// Cooperative launch: the blocks are spread over the whole grid, ONE grid barrier per step; every
// workgroup derives the step's rotations redundantly into its LDS (24 bytes per pair, dynamic:
// jacobi_smem_bytes(nj)), so all take the same exit.
// A ping-pongs between two buffers (step t reads A0/A1, writes the other), so a workgroup still
// deriving its rotations never sees a block another workgroup has already rotated.
// On return evals[j] holds the eigenvalues and row j of QT the eigenvector of evals[j].
#define PMF_NNDSVD_MAX_N 4096          // the full Jacobi decomposition up to here ...
#define PMF_TOPK_MAX_N 16384           // ... the k largest pairs by filtered subspace iteration (pmf_topk.h) up to here
static inline size_t jacobi_smem_bytes(int nj) { return (size_t)(nj / 2) * 24; }

__global__ __launch_bounds__(1024) void k_jacobi_eigh(double* A0, double* A1, double* QT,
                                                      int ld, int nj, int max_sweeps,
                                                      double* __restrict__ evals,
                                                      int* __restrict__ sweeps_done) {
  double* A = A0;       // current
  double* An = A1;      // next
  cooperative_groups::grid_group grid = cooperative_groups::this_grid();
  extern __shared__ double jac_lds[];
  __shared__ int s_rot;
  __shared__ double s_trace;
  const int tid = threadIdx.x;
  const int gtid = blockIdx.x * 1024 + tid, gsize = gridDim.x * 1024;
  const int half = nj >> 1;
  double* sc = jac_lds;
  double* ss = sc + half;
  int* sp = reinterpret_cast<int*>(ss + half);
  int* sq = sp + half;
  for (int idx = gtid; idx < nj * nj; idx += gsize) {
    const int r = idx / nj, c = idx % nj;
    if (r < c) {
      const double v = 0.5 * (A[(int64_t)r * ld + c] + A[(int64_t)c * ld + r]);
      A[(int64_t)r * ld + c] = v;
      A[(int64_t)c * ld + r] = v;
    }
    QT[(int64_t)r * ld + c] = (r == c) ? 1.0 : 0.0;
  }
  grid.sync();
  if (tid == 0) {
    double t = 0.0;
    for (int j = 0; j < nj; ++j) t += fabs(A[(int64_t)j * ld + j]);   // invariant under the rotations
    s_trace = t;
  }
  __syncthreads();
  const double conv = 1e-14 * s_trace / (double)nj;   // a whole sweep below this: converged
  int sweep = 0;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid == 0) s_rot = 0;
    __syncthreads();
    for (int t = 0; t < nj - 1; ++t) {
      for (int pr = tid; pr < half; pr += 1024) {
        int p, q;
        if (pr == 0) { p = nj - 1; q = t; }
        else { p = (t + pr) % (nj - 1); q = (t + nj - 1 - pr) % (nj - 1); }
        if (p > q) { const int x = p; p = q; q = x; }
        const double app = A[(int64_t)p * ld + p], aqq = A[(int64_t)q * ld + q], apq = A[(int64_t)p * ld + q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > 1e-300 && fabs(apq) > 1e-17 * sqrt(fabs(app * aqq))) {
          const double tau = (aqq - app) / (2.0 * apq);
          const double tt = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + tt * tt);
          s = tt * c;
          if (fabs(apq) > conv) s_rot = 1;
        }
        sc[pr] = c; ss[pr] = s; sp[pr] = p; sq[pr] = q;
      }
      __syncthreads();
      for (int idx = gtid; idx < half * half; idx += gsize) {       // 2 x 2 blocks, pr <= pc
        const int pr = idx / half, pc = idx % half;
        if (pr > pc) continue;
        const double s1 = ss[pr], s2 = ss[pc];
        const double c1 = sc[pr], c2 = sc[pc];
        const int64_t p = sp[pr], q = sq[pr], pp = sp[pc], qq = sq[pc];
        const double a = A[p * ld + pp], b = A[p * ld + qq], c = A[q * ld + pp], d = A[q * ld + qq];
        double a2 = a, b2 = b, c2v = c, d2 = d;
        if (s1 != 0.0 || s2 != 0.0) {
          const double a1 = c1 * a - s1 * c, c1v = s1 * a + c1 * c;
          const double b1 = c1 * b - s1 * d, d1 = s1 * b + c1 * d;
          a2 = c2 * a1 - s2 * b1;
          b2 = s2 * a1 + c2 * b1;
          c2v = c2 * c1v - s2 * d1;
          d2 = s2 * c1v + c2 * d1;
          if (pr == pc) { b2 = 0.0; c2v = 0.0; }                     // the annihilated pair, exactly
        }
        An[p * ld + pp] = a2; An[p * ld + qq] = b2; An[q * ld + pp] = c2v; An[q * ld + qq] = d2;
        if (pr != pc) { An[pp * ld + p] = a2; An[qq * ld + p] = b2; An[pp * ld + q] = c2v; An[qq * ld + q] = d2; }
      }
      for (int idx = gtid; idx < half * nj; idx += gsize) {         // rows p, q of QT
        const int pr = idx / nj, col = idx % nj;
        const double s = ss[pr];
        if (s == 0.0) continue;
        const double c = sc[pr];
        double* xp = QT + (int64_t)sp[pr] * ld + col;
        double* yp = QT + (int64_t)sq[pr] * ld + col;
        const double x = *xp, y = *yp;
        *xp = c * x - s * y;
        *yp = s * x + c * y;
      }
      grid.sync();
      { double* x = A; A = An; An = x; }
    }
    const int rot = s_rot;
    __syncthreads();
    if (!rot) { ++sweep; break; }
  }
  for (int j = gtid; j < nj; j += gsize) evals[j] = A[(int64_t)j * ld + j];
  if (gtid == 0) *sweeps_done = sweep;
}

// The k largest eigenvalues (descending; only those > 1e-8 count, svd.py:130-131): order[i] = row
// of QT, sv[i] = sqrt(lambda_i), basis row i = v_i / sv[i] (float32, [KP][np], zero padded) for the
// U = V (v_i / s_i) product.  found[0] = how many of the k exceeded the threshold.
__global__ __launch_bounds__(1024) void k_nndsvd_select(const double* __restrict__ evals,
                                                        const double* __restrict__ QT, int ld, int nj,
                                                        int n, int k, int KP, int np,
                                                        float* __restrict__ B, double* __restrict__ sv,
                                                        int* __restrict__ order, int* __restrict__ found) {
  __shared__ double sval[1024];
  __shared__ int sidx[1024];
  __shared__ double ssv[1024];
  __shared__ int sord[1024];
  const int tid = threadIdx.x;
  constexpr int PER = PMF_NNDSVD_MAX_N / 1024;      // eigenvalues tid, tid + 1024, ... per thread
  double mine[PER];
  unsigned taken = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int e = tid + j * 1024;
    mine[j] = e < nj ? evals[e] : -1.0e300;
    if (e >= nj) taken |= 1u << j;
  }
  int nfound = 0;
  for (int i = 0; i < k; ++i) {
    double best = -1.0e300;
    int bidx = tid;
#pragma unroll
    for (int j = 0; j < PER; ++j)                     // ascending index: ties keep the lowest
      if (!((taken >> j) & 1u) && mine[j] > best) { best = mine[j]; bidx = tid + j * 1024; }
    sval[tid] = best;
    sidx[tid] = bidx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (tid < o) {
        const double a = sval[tid], b = sval[tid + o];
        const int ia = sidx[tid], ib = sidx[tid + o];
        if (b > a || (b == a && ib < ia)) { sval[tid] = b; sidx[tid] = ib; }
      }
      __syncthreads();
    }
    const int win = sidx[0];
    const double val = sval[0];
    if ((win & 1023) == tid) taken |= 1u << (win >> 10);
    const bool ok = val > 1e-8;
    if (ok) ++nfound;
    if (tid == 0) { sord[i] = win; ssv[i] = ok ? sqrt(val) : 0.0; }
    __syncthreads();
  }
  if (tid == 0) found[0] = nfound;
  for (int i = tid; i < KP; i += 1024) { sv[i] = i < k ? ssv[i] : 0.0; order[i] = i < k ? sord[i] : 0; }
  for (int idx = tid; idx < KP * np; idx += 1024) {
    const int i = idx / np, c = idx % np;
    float b = 0.f;
    if (i < k && c < n && ssv[i] > 0.0) b = (float)(QT[(int64_t)sord[i] * ld + c] / ssv[i]);
    B[idx] = b;
  }
}

// part[blk][0][col] = sum of max(u,0)^2, part[blk][1][col] = sum of max(-u,0)^2 over the block's
// rows of U ([.][KP], KP in {16,32,64} or a multiple of 128; blockIdx.y picks the group of 256 columns); float64.
__global__ __launch_bounds__(256) void k_split_norms(const float* __restrict__ U, int64_t m, int KP,
                                                     int64_t rows_per_blk, double* __restrict__ part) {
  __shared__ double sh[2][256];
  const int tid = threadIdx.x;
  const int CW = KP < 256 ? KP : 256;
  const int lc = tid % CW, col = blockIdx.y * CW + lc, rs = tid / CW, nrs = 256 / CW;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  int64_t r1 = r0 + rows_per_blk;
  if (r1 > m) r1 = m;
  double p = 0.0, q = 0.0;
  for (int64_t r = r0 + rs; r < r1 && col < KP; r += nrs) {
    const double u = (double)U[r * KP + col];
    if (u > 0.0) p += u * u; else q += u * u;
  }
  sh[0][tid] = p;
  sh[1][tid] = q;
  __syncthreads();
  if (tid < CW && col < KP) {
    double a = 0.0, b = 0.0;
    for (int j = 0; j < nrs; ++j) { a += sh[0][tid + j * CW]; b += sh[1][tid + j * CW]; }
    part[((int64_t)blockIdx.x * 2 + 0) * KP + col] = a;
    part[((int64_t)blockIdx.x * 2 + 1) * KP + col] = b;
  }
}

__global__ void k_split_sum(const double* __restrict__ part, int nblk, int KP, double* __restrict__ norms) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // [2][KP]
  if (e >= 2 * KP) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += part[(int64_t)b * 2 * KP + e];
  norms[e] = s;
}

__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh /*[2][16]*/) {
  const int tid = threadIdx.x;
  a = wave_sum_f64(a);
  b = wave_sum_f64(b);
  __syncthreads();
  if ((tid & 63) == 0) { sh[tid >> 6] = a; sh[16 + (tid >> 6)] = b; }
  __syncthreads();
  double x = 0.0, y = 0.0;
  for (int w = 0; w < 16; ++w) { x += sh[w]; y += sh[16 + w]; }
  a = x; b = y;
}

// Closed form of nndsvd.py:84-106.  Basis 0: sqrt(s_0) |u_0|, sqrt(s_0) |v_0| (:86,89).  Basis i:
// the larger of |u+||v+| and |u-||v-| picks the sign branch; sigma = s_i * that product;
// W[:,i] = sqrt(sigma) u(+-) / |u(+-)|, H[i,:] = sqrt(sigma) v(+-) / |v(+-)| (:103,106).
// wscale / wmode (0 abs, +1 positive part, -1 negative part) drive k_nndsvd_w.
__global__ __launch_bounds__(1024) void k_nndsvd_finalize(const double* __restrict__ QT, int ld,
                                                          const int* __restrict__ order,
                                                          const double* __restrict__ sv,
                                                          const double* __restrict__ norms, int n, int k,
                                                          int KP, int np, float* __restrict__ H,
                                                          float* __restrict__ wscale, int* __restrict__ wmode) {
  __shared__ double sh[32];
  const int tid = threadIdx.x;
  for (int i = 0; i < KP; ++i) {
    if (i >= k) {
      for (int c = tid; c < np; c += 1024) H[(int64_t)i * np + c] = 0.f;
      if (tid == 0) { wscale[i] = 0.f; wmode[i] = 0; }
      continue;
    }
    const double* qrow = QT + (int64_t)order[i] * ld;
    double vp2 = 0.0, vn2 = 0.0;
    for (int c = tid; c < n; c += 1024) {
      const double v = qrow[c];
      if (v > 0.0) vp2 += v * v; else vn2 += v * v;
    }
    block_sum2(vp2, vn2, sh);
    const double s = sv[i];
    double hp = 0.0, hn = 0.0, ws = 0.0;             // h = hp * max(v,0) + hn * max(-v,0)
    int mode = 0;
    if (i == 0) {
      ws = sqrt(s);
      hp = hn = ws;                                  // sqrt(s_0) |v_0|
    } else {
      const double up = sqrt(norms[i]), un = sqrt(norms[KP + i]);
      const double vp = sqrt(vp2), vn = sqrt(vn2);
      const double a = up * vp, b = un * vn;
      if (a >= b) {
        mode = 1;
        if (a > 0.0) { const double rs = sqrt(s * a); ws = rs / up; hp = rs / vp; }
      } else {
        mode = -1;
        const double rs = sqrt(s * b);
        ws = rs / un;
        hn = rs / vn;
      }
    }
    for (int c = tid; c < np; c += 1024) {
      double h = 0.0;
      if (c < n) { const double v = qrow[c]; h = v > 0.0 ? hp * v : hn * -v; }
      H[(int64_t)i * np + c] = (float)h;
    }
    if (tid == 0) { wscale[i] = (float)ws; wmode[i] = mode; }
  }
}

// W <- the scaled non-negative part of U, in place ([mp][KP]; rows >= m and columns >= k stay 0).
__global__ __launch_bounds__(256) void k_nndsvd_w(float* __restrict__ W, int64_t total, int KP, int64_t m,
                                                  const float* __restrict__ wscale,
                                                  const int* __restrict__ wmode) {
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {   // (grid-stride: > 2^32 elements)
    const int64_t r = idx / KP;
    const int col = (int)(idx % KP);
    const float u = W[idx], s = wscale[col];
    const int mode = wmode[col];
    float o = 0.f;
    if (r < m) o = mode == 0 ? s * fabsf(u) : (mode > 0 ? s * fmaxf(u, 0.f) : s * fmaxf(-u, 0.f));
    W[idx] = o;
  }
}
