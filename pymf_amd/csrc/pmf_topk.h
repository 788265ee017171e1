// pmf_topk.h -- the k largest eigenpairs of the symmetric positive semi-definite Gram matrix A = data^T data for the NNDSVD
// initialisation (pymf/svd.py:125-148 takes ALL n of them from scipy.linalg.eigh and keeps the leading ones) when n is
// beyond the full Jacobi decomposition of pmf_nndsvd.h (13 s at n = 4096, O(n^3)).
//
// Chebyshev-filtered subspace iteration on a block of s = k + p row vectors (Y is [s][ld], row j = vector j, float64):
//   filter     Y <- T_d((A' - c) / e) Y, the three-term recurrence; [0, cut] is damped, cut = the smallest Ritz value of the
//              block, and the degree d is chosen so that the growth of the largest active direction over the k-th stays
//              below 1e9 (a spectrum with one dominant eigenvalue -- every non-negative data matrix has one -- would
//              otherwise swamp the rest of the block in rounding);
//   ortho      rows against the locked vectors (twice), then among themselves through the eigen-decomposition of their
//              s x s Gram matrix (k_jacobi_eigh), twice;
//   Rayleigh-Ritz   T = Y A' Y^T, eigh(T) on the device (k_jacobi_eigh), rows rotated, residuals ||A y - theta y||;
//   lock       the leading Ritz pairs whose residual is below 1e-13 lambda_1, IN ORDER; locked pairs are deflated
//              implicitly (A' y = A y - L^T diag(theta) L y), the block is refilled with random rows.
// The products are float64 MFMA (tile_dgemm of pmf_inv.h, one wave per 16 x 16 tile); the control flow (degree, interval,
// lock count) runs on the host, one synchronisation per Rayleigh-Ritz step.  A numpy model of exactly this loop
// (uniform, low-rank + noise, decaying and rank-deficient spectra) converges in 50-180 products with A to eigenvectors
// within 3e-15 of LAPACK's.
#pragma once
#include "pmf_dev.h"

// A <- (A + A^T) / 2 over the upper / lower pairs (the Gram matrix arrives as column-block passes of fp32 MFMA products:
// its two triangles differ in the last float32 digits; the Jacobi path symmetrises the same way, pmf_nndsvd.h)
__global__ __launch_bounds__(256) void k_topk_symmetrise(double* __restrict__ A, int ld, int n) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)n * n) return;
  const int r = (int)(q / n), c = (int)(q % n);
  if (r < c) {
    const double v = 0.5 * (A[(int64_t)r * ld + c] + A[(int64_t)c * ld + r]);
    A[(int64_t)r * ld + c] = v;
    A[(int64_t)c * ld + r] = v;
  }
}

// out = ((Z - D) - cshift * Y1) * alpha - beta * Y0     (D, Y0 may be null)
__global__ __launch_bounds__(256) void k_topk_cheb(const double* __restrict__ Z, const double* __restrict__ D,
                                                   const double* __restrict__ Y1, const double* __restrict__ Y0,
                                                   double* __restrict__ out, int64_t count, double cshift, double alpha,
                                                   double beta) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= count) return;
  double z = Z[q];
  if (D) z -= D[q];
  double v = (z - cshift * Y1[q]) * alpha;
  if (Y0) v -= beta * Y0[q];
  out[q] = v;
}

// Y -= D
__global__ __launch_bounds__(256) void k_topk_sub(double* __restrict__ Y, const double* __restrict__ D, int64_t count) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q < count) Y[q] -= D[q];
}

// C[r][j] *= th[j]  (C is [rows][ld], j < cols)
__global__ __launch_bounds__(256) void k_topk_scale_cols(double* __restrict__ C, int rows, int cols, int ld,
                                                         const double* __restrict__ th) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= rows * cols) return;
  const int r = q / cols, j = q % cols;
  C[(int64_t)r * ld + j] *= th[j];
}

// Y[j][:] *= sc[j]
__global__ __launch_bounds__(256) void k_topk_scale_rows(double* __restrict__ Y, int rows, int ld, const double* __restrict__ sc) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)rows * ld) return;
  Y[q] *= sc[q / ld];
}

// dst[j][:] = src[perm[j]][:]   (width floats of a row; ld_src / ld_dst leading dimensions)
__global__ __launch_bounds__(256) void k_topk_gather_rows(const double* __restrict__ src, int64_t ld_src, const int* __restrict__ perm,
                                                          double* __restrict__ dst, int64_t ld_dst, int rows, int width) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)rows * width) return;
  const int j = (int)(q / width), c = (int)(q % width);
  dst[(int64_t)j * ld_dst + c] = src[(int64_t)perm[j] * ld_src + c];
}

// out[j] = || GQ[j][:] - theta[j] Q[j][:] ||_2 ; one workgroup per row
__global__ __launch_bounds__(256) void k_topk_resid(const double* __restrict__ GQ, const double* __restrict__ Q, int ld, int n,
                                                    const double* __restrict__ theta, double* __restrict__ out) {
  __shared__ double sh[256];
  const int j = blockIdx.x, tid = threadIdx.x;
  const double th = theta[j];
  double s = 0.0;
  for (int c = tid; c < n; c += 256) {
    const double r = GQ[(int64_t)j * ld + c] - th * Q[(int64_t)j * ld + c];
    s = fma(r, r, s);
  }
  sh[tid] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sh[tid] += sh[tid + o];
    __syncthreads();
  }
  if (tid == 0) out[j] = sqrt(sh[0]);
}

// rows [r0, r1) of Y: uniform (-1, 1) in columns < n, 0 in the padding (A's padding rows / columns are 0: it stays 0)
__global__ __launch_bounds__(256) void k_topk_fill_random(double* __restrict__ Y, int r0, int r1, int ld, int n, uint64_t seed) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= (int64_t)(r1 - r0) * ld) return;
  const int r = r0 + (int)(q / ld), c = (int)(q % ld);
  Y[(int64_t)r * ld + c] = c < n ? 2.0 * (double)u01_from(seed, (uint64_t)r * (uint64_t)ld + c) - 1.0 : 0.0;
}
