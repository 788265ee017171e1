// pmf_csr.h -- SNMF on scipy.sparse CSR data (BASELINE cfg5).  The reference cannot run
// SNMF on sparse input; the semantics here are dense SNMF on V.toarray():
//   update_w (snmf.py:67-70):  W = (V H^T) inv(H H^T) = V (H^T inv(H H^T)) = V M,
//       M (n x k) is formed once per step by a small dense product, so the sparse side
//       is one SpMM pass that writes W once and never materialises V H^T;
//   update_h (snmf.py:79):     XW^T = W^T V accumulated per row chunk in LDS (transposed,
//       lanes <-> bases so LDS adds are conflict-free) and reduced like the dense slabs.
#pragma once
#include "pmf_dev.h"

// M[col][kk'] = sum_kk H[kk][col] * GinvT[kk'][kk]      (GinvT[a][b] = inv[b][a])
__global__ void k_snmf_m(const float* __restrict__ H, int64_t ldh, int n_cols, int KP,
                         const float* __restrict__ GinvT, float* __restrict__ M) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_cols * KP) return;
  const int col = q / KP, kp = q % KP;
  float s = 0.f;
  for (int kk = 0; kk < KP; ++kk) s = fmaf(H[(int64_t)kk * ldh + col], GinvT[kp * KP + kk], s);
  M[(int64_t)col * KP + kp] = s;
}

// W[row][:] = sum over the row's non-zeros of val * M[col][:]; one wave per row, lanes <-> bases.
template <int VPL>   // bases per lane: KP = 64 * VPL or less
__global__ __launch_bounds__(256) void k_csr_w(const int64_t* __restrict__ indptr,
                                               const int32_t* __restrict__ indices,
                                               const float* __restrict__ vals, int64_t rows, int KP,
                                               const float* __restrict__ M, float* __restrict__ W) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int64_t row = w0; row < rows; row += (int64_t)gridDim.x * 4) {
    const int64_t a = indptr[row], b = indptr[row + 1];
    float acc[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) acc[v] = 0.f;
    for (int64_t e = a; e < b; ++e) {
      const int col = indices[e];
      const float val = vals[e];
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int kk = lane + 64 * v;
        if (kk < KP) acc[v] = fmaf(val, M[(int64_t)col * KP + kk], acc[v]);
      }
    }
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int kk = lane + 64 * v;
      if (kk < KP) W[row * KP + kk] = acc[v];
    }
  }
}

// slab[chunk] P part (KP x np, ld = np + KP) = W_chunk^T V_chunk for CSR V.
// LDS: Pt[np][KP] (transposed) so that a non-zero adds a contiguous KP vector.
template <int VPL>
__global__ __launch_bounds__(256) void k_csr_p(const int64_t* __restrict__ indptr,
                                               const int32_t* __restrict__ indices,
                                               const float* __restrict__ vals, int64_t rows,
                                               int rows_per_chunk, int KP, int np,
                                               const float* __restrict__ W, float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) float pt[];   // [np][KP]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int q = tid; q < np * KP; q += 256) pt[q] = 0.f;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  for (int64_t row = r0 + wv; row < r1; row += 4) {
    const int64_t a = indptr[row], b = indptr[row + 1];
    if (a == b) continue;
    float w[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int kk = lane + 64 * v;
      w[v] = kk < KP ? W[row * KP + kk] : 0.f;
    }
    for (int64_t e = a; e < b; ++e) {
      const int col = indices[e];
      const float val = vals[e];
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int kk = lane + 64 * v;
        if (kk < KP) atomicAdd(&pt[col * KP + kk], val * w[v]);
      }
    }
  }
  __syncthreads();
  const int64_t ldp = (int64_t)np + KP;
  float* base = slab + (int64_t)blockIdx.x * KP * ldp;
  for (int q = tid; q < np * KP; q += 256) {
    const int kk = q / np, col = q % np;
    base[(int64_t)kk * ldp + col] = pt[col * KP + kk];
  }
}
