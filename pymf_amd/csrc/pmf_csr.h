// pmf_csr.h -- SNMF on scipy.sparse CSR data (BASELINE cfg5).  The reference cannot run
// SNMF on sparse input; the semantics here are dense SNMF on V.toarray():
//   update_w (snmf.py:67-70):  W = (V H^T) inv(H H^T) = V (H^T inv(H H^T)) = V M,
//       M (n x k) is formed once per step in float64 (k_snmf_mt, pmf_small.h), so the sparse side
//       is one SpMM pass that writes W once and never materialises V H^T;
//   update_h (snmf.py:79):     XW^T = W^T V accumulated per row chunk in LDS (transposed,
//       lanes <-> bases so LDS adds are conflict-free) and reduced like the dense slabs.
#pragma once
#include "pmf_dev.h"

// C = V^T V (np x np, float64) of CSR data, for the Gram-space SNMF loop: a row with c entries adds its
// c^2 products v_a v_b to C[col_a][col_b] (duplicates included: (a + b)^2 = aa + ab + ba + bb, the semantics
// of V.toarray()).  C is symmetric and v_a v_b = v_b v_a exactly, so only the pairs with col_a <= col_b are
// added (to the upper triangle) and k_csr_gram_sum mirrors.
//
// A wave takes 64 ROWS at a time (lane <-> row: two coalesced loads bring the 65 row pointers), stages the
// block's T entries (column, value) and, per entry, the row it belongs to in a small LDS area of its own, and
// then works lane <-> ENTRY: entry t of row r is paired with the c_r entries of its row, one per trip, so a
// trip of the wave issues up to 64 products whatever the row lengths are (1.28 entries per row at cfg5: one
// wave per row, lanes <-> pairs -- the first form of this kernel -- left 62 lanes idle and spent three
// dependent memory latencies per row: 1.48 ms for 77 MB).  The loads are software-pipelined: while block i
// is paired, the entries of block i + 1 and the row pointers of block i + 2 are in flight (a wave owns a
// CONTIGUOUS range of blocks).  A block with GRAM_CAP or more entries (dense rows) takes the row-by-row
// form.  Every workgroup accumulates into a private image of C's upper triangle -- in LDS when it fits (np <= 128), else
// directly into its slab in global memory -- and k_csr_gram_sum adds the slabs.
// Round 6: the accumulation is EXACT FIXED POINT, hence independent of the order in which the waves' atomics land.  Rounds 2-5
// added the (exact) float32 x float32 products with float64 atomics: the order of those additions was the one run-to-run
// freedom of the whole library (1e-16 relative) -- invisible while H was rounded to float32 between the iterations, but with
// SNMF's H in float64 (pmf_inv.h) cond(H H^T) ~ 1e7 carries it into the last bits of W.  With |v| < 2^e for all entries a product
// p (|p| < 2^2e, 48 significant bits) is cut into two signed limbs on the grids u1 = 2^(2e-32) and u2 = 2^(2e-64),
//     hi = trunc(p / u1),   lo = trunc((p - hi u1) / u2),        |hi|, |lo| < 2^32,
// each added with a 64-bit INTEGER atomic (associative: any order, same bits; up to 2^31 summands per entry); what is cut
// off below u2 is a function of the product alone (deterministic) and < 2^-64 of the largest product -- closer to the exact
// sum than the float64 additions were.  k_csr_gram_sum adds the slabs' limbs as integers and converts once.
constexpr int GRAM_CAP = 256;       // entries of a 64-row block staged per wave
constexpr int GRAM_WAVES = 8;       // waves per workgroup
__host__ __device__ constexpr size_t gram_tri(int np) { return (size_t)np * (np + 1) / 2; }          // entries of the upper triangle
__device__ __forceinline__ int gram_tri_index(int c1, int c2, int np) { return c1 * np - c1 * (c1 - 1) / 2 + (c2 - c1); }   // c1 <= c2
struct GramScale { double u1, inv_u1, inv_u2; };     // u1 = 2^(2e-32), its inverse, and 1 / u2 = 2^(64-2e)
__device__ __forceinline__ void gram_add(unsigned long long* img, int idx, double p, const GramScale& g) {
  const double h = trunc(p * g.inv_u1);             // (scalings by powers of two and the difference below are exact)
  const double l = trunc((p - h * g.u1) * g.inv_u2);
  atomicAdd(&img[2 * idx], (unsigned long long)(long long)h);
  atomicAdd(&img[2 * idx + 1], (unsigned long long)(long long)l);
}
// bit pattern of the largest |v| (non-negative floats order like their bit patterns; a NaN sorts above infinity)
__global__ __launch_bounds__(256) void k_absmax_bits_f32(const float* __restrict__ v, int64_t n, unsigned* __restrict__ out) {
  unsigned mx = 0u;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) mx = max(mx, __float_as_uint(fabsf(v[e])));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(out, mx);
}
constexpr size_t gram_stage_bytes() { return (size_t)GRAM_WAVES * (GRAM_CAP * (4 + 4 + 1) + 64 * 4 + 64); }

__global__ __launch_bounds__(64 * GRAM_WAVES) void k_csr_gram(const int64_t* __restrict__ indptr,
                                                              const int32_t* __restrict__ indices,
                                                              const float* __restrict__ vals, int64_t rows, int np,
                                                              unsigned long long* __restrict__ slabs, int use_lds, GramScale gs) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sC[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int E = 2 * (int)gram_tri(np);              // two limbs per entry of the upper triangle
  unsigned long long* img = use_lds ? sC : slabs + (size_t)blockIdx.x * E;
  // per-wave staging behind the image: columns, values, row of every entry, first entry and length of every row
  char* stage = reinterpret_cast<char*>(sC + (use_lds ? E : 0)) + (size_t)wv * (GRAM_CAP * 9 + 64 * 4 + 64);
  int* sCol = reinterpret_cast<int*>(stage);
  float* sVal = reinterpret_cast<float*>(stage + GRAM_CAP * 4);
  int* sRel = reinterpret_cast<int*>(stage + GRAM_CAP * 8);
  unsigned char* sCnt = reinterpret_cast<unsigned char*>(stage + GRAM_CAP * 8 + 64 * 4);
  unsigned char* sRow = reinterpret_cast<unsigned char*>(stage + GRAM_CAP * 8 + 64 * 4 + 64);
  if (use_lds) {
    for (int q = tid; q < E; q += 64 * GRAM_WAVES) sC[q] = 0ull;
    __syncthreads();
  }
  // this wave's contiguous range of 64-row blocks
  const int64_t nblk = (rows + 63) / 64;
  const int64_t nwaves = (int64_t)gridDim.x * GRAM_WAVES, gw = (int64_t)blockIdx.x * GRAM_WAVES + wv;
  const int64_t per = nblk / nwaves, extra = nblk % nwaves;
  const int64_t b0 = gw * per + (gw < extra ? gw : extra), nb = per + (gw < extra ? 1 : 0);
  const int64_t nnz = indptr[rows];

  auto load_ptrs = [&](int64_t blk, int64_t& ip, int64_t& ipn) {      // row pointers of block blk (lane <-> row)
    const int64_t r = blk * 64 + lane;
    ip = indptr[r < rows ? r : rows];
    ipn = indptr[r + 1 < rows ? r + 1 : rows];
  };
  int ecol[GRAM_CAP / 64];
  float eval_[GRAM_CAP / 64];
  auto load_entries = [&](int64_t a0, int T) {                          // entries [a0, a0 + min(T, CAP)) -> registers
#pragma unroll
    for (int u = 0; u < GRAM_CAP / 64; ++u) {
      const int t = 64 * u + lane;
      ecol[u] = 0; eval_[u] = 0.f;
      if (t < T && a0 + t < nnz) { ecol[u] = indices[a0 + t]; eval_[u] = vals[a0 + t]; }
    }
  };
  auto first64 = [&](int64_t v) -> int64_t {                            // lane 0's value, wave-uniform
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
  };

  int64_t ipA = 0, ipnA = 0, ipB = 0, ipnB = 0;
  if (nb > 0) load_ptrs(b0, ipA, ipnA);
  if (nb > 1) load_ptrs(b0 + 1, ipB, ipnB);
  int64_t a0 = 0;
  int T = 0;
  if (nb > 0) {
    a0 = first64(ipA);
    T = (int)(__builtin_amdgcn_readlane((int)(ipnA - a0), 63));
    load_entries(a0, T);
  }
  for (int64_t bi = 0; bi < nb; ++bi) {
    const int rel = (int)(ipA - a0), cnt = (int)(ipnA - ipA);
    const int Tcur = T;
    const int64_t a0cur = a0;
    const bool staged = Tcur < GRAM_CAP;          // (row lengths are staged as bytes: < 256 entries in all)
    if (staged) {
      // stage block bi (its entries are in registers), then request block bi + 1's entries and bi + 2's pointers
#pragma unroll
      for (int u = 0; u < GRAM_CAP / 64; ++u) {
        const int t = 64 * u + lane;
        if (t < Tcur) { sCol[t] = ecol[u]; sVal[t] = eval_[u]; }
      }
      sRel[lane] = rel;
      sCnt[lane] = (unsigned char)(cnt < 255 ? cnt : 255);
      for (int j = 0; j < cnt; ++j) sRow[rel + j] = (unsigned char)lane;
    }
    // pipeline: next block's pointers are at hand (ipB), its entries are requested now
    int64_t ipC = 0, ipnC = 0;
    if (bi + 2 < nb) load_ptrs(b0 + bi + 2, ipC, ipnC);
    int64_t a0n = 0;
    int Tn = 0;
    if (bi + 1 < nb) {
      a0n = first64(ipB);
      Tn = (int)(__builtin_amdgcn_readlane((int)(ipnB - a0n), 63));
      if (staged) load_entries(a0n, Tn);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the wave's own LDS writes are in place
    if (staged) {
      for (int tb = 0; tb < Tcur; tb += 64) {
        const int t = tb + lane;
        const bool act = t < Tcur;
        const int r = act ? sRow[t] : 0;
        const int ea = sRel[r];
        const int c = act ? sCnt[r] : 0;
        const int col1 = act ? sCol[t] : 0;
        const double v1 = act ? (double)sVal[t] : 0.0;
        int cm = c;                                                     // longest row among this trip's entries
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) cm = max(cm, __shfl_xor(cm, o, 64));
        for (int j = 0; j < cm; ++j) {
          if (j < c) {
            const int col2 = sCol[ea + j];
            if (col1 <= col2) gram_add(img, gram_tri_index(col1, col2, np), v1 * (double)sVal[ea + j], gs);
          }
        }
      }
    } else {
      // a block of long rows: row by row, lanes <-> pairs of entries (reads straight from global)
      for (int rr = 0; rr < 64; ++rr) {
        const int ra = __builtin_amdgcn_readlane(rel, rr), rc = __builtin_amdgcn_readlane(cnt, rr);
        const int64_t base = a0cur + ra;
        for (int64_t pq = lane; pq < (int64_t)rc * rc; pq += 64) {
          const int e1 = (int)(pq / rc), e2 = (int)(pq % rc);
          const int col1 = indices[base + e1], col2 = indices[base + e2];
          if (col1 <= col2) gram_add(img, gram_tri_index(col1, col2, np), (double)vals[base + e1] * (double)vals[base + e2], gs);
        }
      }
      if (bi + 1 < nb) load_entries(a0n, Tn);
    }
    __builtin_amdgcn_wave_barrier();
    ipA = ipB; ipnA = ipnB; ipB = ipC; ipnB = ipnC;
    a0 = a0n; T = Tn;
  }
  if (use_lds) {
    __syncthreads();
    unsigned long long* out = slabs + (size_t)blockIdx.x * E;
    for (int q = tid; q < E; q += 64 * GRAM_WAVES) out[q] = sC[q];
  }
}

// C = sum of the slabs' upper triangles, mirrored.  Block b owns 64 consecutive elements of the np x np matrix; thread
// (g = tid / 64, e = tid % 64) adds the limbs of slabs g, g + 4, ... as INTEGERS (independent loads, eight in flight; any
// order gives the same bits), the four partial sums likewise, and the two 64-bit sums become one float64: their 32-bit halves
// convert exactly and are added from the smallest grid up.  finite == 0: the data held a NaN or an infinity -- C is NaN.
__global__ __launch_bounds__(256) void k_csr_gram_sum(const unsigned long long* __restrict__ slabs, int nslabs, int np,
                                                      double* __restrict__ C, GramScale gs, int finite) {
  __shared__ long long part[4][64][2];
  const int64_t E = (int64_t)np * np;
  const size_t T2 = 2 * gram_tri(np);
  const int g = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int r = (int)(e / np), c = (int)(e % np);
  const bool live = e < E && r <= c;
  long long sh = 0, sl = 0;
  if (live) {
    const size_t t = 2 * (size_t)gram_tri_index(r, c, np);
#pragma unroll 8
    for (int q = g; q < nslabs; q += 4) { sh += (long long)slabs[(size_t)q * T2 + t]; sl += (long long)slabs[(size_t)q * T2 + t + 1]; }
  }
  part[g][threadIdx.x & 63][0] = sh;
  part[g][threadIdx.x & 63][1] = sl;
  __syncthreads();
  if (g == 0 && live) {
    const int l = threadIdx.x & 63;
    const long long H = part[0][l][0] + part[1][l][0] + part[2][l][0] + part[3][l][0];
    const long long L = part[0][l][1] + part[1][l][1] + part[2][l][1] + part[3][l][1];
    // X = Xa 2^32 + Xb with Xb in [0, 2^32): both halves are exact as doubles
    const double Ha = (double)(H >> 32), Hb = (double)(unsigned)(H & 0xffffffffll);
    const double La = (double)(L >> 32), Lb = (double)(unsigned)(L & 0xffffffffll);
    const double u1 = gs.u1, u2 = 1.0 / gs.inv_u2;
    double t = ((Lb * u2 + La * (u2 * 4294967296.0)) + Hb * u1) + Ha * (u1 * 4294967296.0);
    if (!finite) t = __builtin_nan("");
    C[e] = t;
    C[(int64_t)c * np + r] = t;
  }
}

// W = V M for CSR V, the write-bound form (the Gram-space SNMF loop materialises W with it, once per
// factorize()).  A wave takes 16-row blocks: ONE load brings the block's 17 row pointers, one (or more)
// its column/value run; then 64 / (KP / 4) rows are formed at a time, KP / 4 lanes per row with four
// bases each, so every store is a 16-byte-per-lane, 1-KiB-per-instruction contiguous piece of W.  M
// (np x KP) sits in LDS when it fits (64 KiB at n = k = 128: conflict-free float4 row reads) and is read
// through L2 otherwise.  1024-thread workgroups: 16 waves share one M image, 256 rows in flight per CU.
// Round 4: for matrices many workgroups deep the launch is NOT persistent -- one workgroup per 256 rows, M through L2
// (m_in_lds = 0, no image to stage): each workgroup writes one contiguous 128 KiB piece of W and leaves, the pieces
// following each other through memory in dispatch order: 6.75 TB/s at cfg5 against 5.4 for 512 persistent workgroups
// striding through W (tools/csrw_lab.hip; pure store streams of the two shapes: 5.9-6.5 / 5.1-5.6 TB/s).
template <int NT>
__global__ __launch_bounds__(1024) void k_csr_w_blocks(const int64_t* __restrict__ indptr,
                                                       const int32_t* __restrict__ indices,
                                                       const float* __restrict__ vals, int64_t nblk, int np,
                                                       const float* __restrict__ M, float* __restrict__ W,
                                                       int m_in_lds) {
  constexpr int KP = 16 * NT, LPR = KP / 4, RPI = 64 / LPR;     // lanes per row, rows per store instruction
  extern __shared__ __attribute__((aligned(16))) float sMw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (m_in_lds) {
    for (int q = tid; q < np * KP / 4; q += 1024) reinterpret_cast<f32x4*>(sMw)[q] = reinterpret_cast<const f32x4*>(M)[q];
    __syncthreads();
  }
  const float* Msrc = m_in_lds ? sMw : M;
  const int sub = lane / LPR, c4 = 4 * (lane % LPR);
  const int64_t nwaves = (int64_t)gridDim.x * 16;
  for (int64_t blk = (int64_t)blockIdx.x * 16 + wv; blk < nblk; blk += nwaves) {
    const int64_t r0 = blk * 16;
    const long long ipv = (long long)indptr[r0 + (lane < 17 ? lane : 16)];
    const unsigned alo = (unsigned)__builtin_amdgcn_readlane((int)(ipv & 0xffffffffll), 0);
    const unsigned ahi = (unsigned)__builtin_amdgcn_readlane((int)(ipv >> 32), 0);
    const long long a = (long long)(((unsigned long long)ahi << 32) | alo);
    const int rel = (int)(ipv - a);                               // lane r < 17: first entry of row r, block-relative
    const int nzb = __builtin_amdgcn_readlane(rel, 16);
    int colv = 0;
    float valv = 0.f;
    if (lane < nzb) { colv = indices[a + lane]; valv = vals[a + lane]; }
#pragma unroll
    for (int g = 0; g < 16 / RPI; ++g) {
      const int row = g * RPI + sub;
      const int ea = __shfl(rel, row, 64), eb = __shfl(rel, row + 1, 64);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int cnt = eb - ea;
      int cmax = cnt;                                             // wave-uniform trip count: longest row of the group
#pragma unroll
      for (int o = 32; o >= LPR; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o, 64));
      for (int it = 0; it < cmax; ++it) {
        const int e = ea + it;
        int col = __shfl(colv, e & 63, 64);
        float val = __shfl(valv, e & 63, 64);
        if (it < cnt && e >= 64) { col = indices[a + e]; val = vals[a + e]; }
        if (it >= cnt) { col = 0; val = 0.f; }
        const f32x4 mrow = *reinterpret_cast<const f32x4*>(Msrc + (size_t)col * KP + c4);
        acc += val * mrow;                                        // duplicates add up, as in V.toarray()
      }
#ifdef PMF_CSRW_PLAIN
      *reinterpret_cast<f32x4*>(W + (size_t)(r0 + row) * KP + c4) = acc;
#else
      __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(W + (size_t)(r0 + row) * KP + c4));   // W is written once, never re-read here
#endif
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

// ---------------------------------------------------------------------------------------------
// One pass over the CSR rows per SNMF iteration (cfg5): per 16-row block a wave
//   (1) forms the new W rows  W[r][:] = sum_nz val * M[col][:]   (M = H^T inv(H H^T) in LDS),
//       writes them to HBM once and into its LDS tile,
//   (2) scatters  Pt[col][:] += val * W[r][:]   (P = W^T V, transposed LDS accumulator shared by
//       the workgroup, LDS float atomics),
//   (3) accumulates S += W_b^T W_b on MFMA from the LDS tile (tiles on/above the diagonal only).
// W is never re-read: HBM traffic = CSR arrays once + W written once.  One workgroup per CU
// (M 4*np*KP B + Pt 4*np*KP B + 4 tiles of 16 x KP floats of LDS).
// Output: row-major slab[blockIdx] = (P | S) like k_colgemm, reduced by k_reduce_slabs.
template <int NT>
__global__ __launch_bounds__(256, 1) void k_snmf_csr_fused(const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices,
                                                           const float* __restrict__ vals,
                                                           int blk_per, int blk_extra, int np,
                                                           const float* __restrict__ M,
                                                           float* __restrict__ W,
                                                           float* __restrict__ slab) {
  constexpr int KP = 16 * NT;
  constexpr int VPL = (KP + 63) / 64;
  constexpr int NS = NT * (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sM = sm;                          // [np][KP]
  float* sPt = sM + (size_t)np * KP;       // [np][KP]
  float* sWall = sPt + (size_t)np * KP;    // 4 x [16][KP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  float* sWt = sWall + wv * (16 * KP);
  for (int q = tid; q < np * KP / 4; q += 256) {
    reinterpret_cast<f32x4*>(sM)[q] = reinterpret_cast<const f32x4*>(M)[q];
    reinterpret_cast<f32x4*>(sPt)[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();

  const int gw = blockIdx.x * 4 + wv;
  const int b0 = gw * blk_per + (gw < blk_extra ? gw : blk_extra);
  const int nb = blk_per + (gw < blk_extra ? 1 : 0);

  f32x4 S[NS];
#pragma unroll
  for (int t = 0; t < NS; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline over blocks: the row pointers of block b+1 are requested while block b's rows
  // are processed, its column/value run while block b's MFMAs execute -- the two dependent HBM
  // round trips per block are off the critical path.
  auto load_ip = [&](int blk) -> long long {
    return (long long)indptr[(int64_t)blk * 16 + (lane < 17 ? lane : 16)];
  };
  auto base_of = [&](long long ipv) -> long long {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(ipv & 0xffffffffll), 0);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(ipv >> 32), 0);
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  long long ipv = nb > 0 ? load_ip(b0) : 0;
  long long a = base_of(ipv);
  int rel = (int)(ipv - a);
  int nzb = __builtin_amdgcn_readlane(rel, 16);
  int colv = 0;
  float valv = 0.f;
  if (lane < nzb) { colv = indices[a + lane]; valv = vals[a + lane]; }

  for (int b = 0; b < nb; ++b) {
    const int64_t r0 = (int64_t)(b0 + b) * 16;
    const bool more = b + 1 < nb;
    const long long ipv_n = more ? load_ip(b0 + b + 1) : 0;       // in flight during the row loop
    auto fetch = [&](int e, int& col, float& val) {       // e wave-uniform
      if (e < 64) {
        col = __builtin_amdgcn_readlane(colv, e);
        val = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(valv), e));
      } else {
        col = indices[a + e];
        val = vals[a + e];
      }
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ea_u = __builtin_amdgcn_readlane(rel, r), eb_u = __builtin_amdgcn_readlane(rel, r + 1);
      float acc[VPL];
#pragma unroll
      for (int v = 0; v < VPL; ++v) acc[v] = 0.f;
      for (int e = ea_u; e < eb_u; ++e) {
        int col; float val;
        fetch(e, col, val);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int kk = lane + 64 * v;
          if (kk < KP) acc[v] = fmaf(val, sM[col * KP + kk], acc[v]);
        }
      }
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        const int kk = lane + 64 * v;
        if (kk < KP) {
          W[(r0 + r) * KP + kk] = acc[v];
          sWt[r * KP + kk] = acc[v];
        }
      }
      for (int e = ea_u; e < eb_u; ++e) {          // P^T[col][:] += val * W[r][:]
        int col; float val;
        fetch(e, col, val);
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
          const int kk = lane + 64 * v;
          if (kk < KP) atomicAdd(&sPt[col * KP + kk], val * acc[v]);
        }
      }
    }
    // next block's column/value run: requested now, lands under the MFMAs below
    const long long a_n = base_of(ipv_n);
    const int rel_n = (int)(ipv_n - a_n);
    const int nzb_n = __builtin_amdgcn_readlane(rel_n, 16);
    int colv_n = 0;
    float valv_n = 0.f;
    if (more && lane < nzb_n) { colv_n = indices[a_n + lane]; valv_n = vals[a_n + lane]; }
    // S += W_b^T W_b: A[i = base][k = row]; MFMA step j contracts rows {4q + j}
    float af[NT][4];
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) af[mt][j] = sWt[(4 * kq + j) * KP + 16 * mt + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int t = 0;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = mt; nt < NT; ++nt, ++t) S[t] = mfma16(af[mt][j], af[nt][j], S[t]);
    }
    ipv = ipv_n; a = a_n; rel = rel_n; nzb = nzb_n; colv = colv_n; valv = valv_n;
  }

  // ---- S: sum the 4 waves through LDS (the M image is dead now), wave 0 writes it row-major ----
  __syncthreads();
  f32x4* ex = reinterpret_cast<f32x4*>(sM);     // NS*64 f32x4 = NS KiB per region, 2 regions <= 4*np*KP B
  auto put = [&](int region) {
#pragma unroll
    for (int t = 0; t < NS; ++t) ex[((size_t)region * NS + t) * 64 + lane] = S[t];
  };
  auto add = [&](int region) {
#pragma unroll
    for (int t = 0; t < NS; ++t) S[t] += ex[((size_t)region * NS + t) * 64 + lane];
  };
  const bool two_regions = (size_t)2 * NS * 1024 <= (size_t)np * KP * 4;
  if (two_regions) {
    if (wv >= 2) put(wv - 2);
    __syncthreads();
    if (wv < 2) add(wv);
    __syncthreads();
    if (wv == 1) put(0);
    __syncthreads();
    if (wv == 0) add(0);
  } else {                                        // tiny n*k: one region, three rounds
    for (int src = 1; src < 4; ++src) {
      if (wv == src) put(0);
      __syncthreads();
      if (wv == 0) add(0);
      __syncthreads();
    }
  }
  const int64_t ldp = (int64_t)np + KP;
  float* base = slab + (int64_t)blockIdx.x * KP * ldp;
  if (wv == 0) {
    int t = 0;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int nt = mt; nt < NT; ++nt, ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) base[(int64_t)(16 * mt + 4 * kq + j) * ldp + np + 16 * nt + i] = S[t][j];
        if (nt > mt)
          *reinterpret_cast<f32x4*>(base + (int64_t)(16 * nt + i) * ldp + np + 16 * mt + 4 * kq) = S[t];
      }
  }
  // ---- P = Pt^T (all waves; Pt is complete since the barrier above) ----
  for (int q = tid; q < np * KP; q += 256) {
    const int kk = q / np, col = q % np;
    base[(int64_t)kk * ldp + col] = sPt[col * KP + kk];
  }
}

// ---------------------------------------------------------------------------------------------
// k_snmf_csr_mfma<NT,NTP>: the one-pass CSR iteration with P = W^T V on MFMA as well.
// Per 16-row block a wave
//   (1) scatters the block's non-zeros into its private, otherwise all-zero, dense 16 x np LDS tile
//       (one ds_add per non-zero, all lanes at once) and ORs a bit per occupied (row%4, column
//       tile) into a 32-bit occupancy mask;
//   (2) forms the new W rows directly in the MFMA A-fragment layout: lane (i, q) accumulates
//       W[4q + j][16*mt + i], j = 0..3, from M = H^T inv(H H^T) in LDS -- 4 rows in flight per step;
//   (3) S += W_b^T W_b (upper tiles) and, only for occupied (j, column tile) pairs,
//       P[:, tile] += W_b^T V_b on MFMA from the dense tile; about half of the pairs are empty at 1 %;
//   (4) writes zeros back over the scattered entries (the tile is all-zero again).
// No float atomics on P, fixed summation order => reproducible.  P (NT x NTP tiles) and S stay
// in registers for the wave's row range; one workgroup per CU.
template <int NT, int NTP>
__global__ __launch_bounds__(256, 1) void k_snmf_csr_mfma(const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ indices,
                                                          const float* __restrict__ vals,
                                                          int blk_per, int blk_extra,
                                                          const float* __restrict__ M,
                                                          float* __restrict__ W,
                                                          float* __restrict__ slab) {
  constexpr int KP = 16 * NT, NP = 16 * NTP;
  constexpr int NS = NT * (NT + 1) / 2;
  static_assert(4 * NTP <= 32, "occupancy mask is 32 bits");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sM = sm;                                   // [NP][KP]
  float* sVall = sM + NP * KP;                      // 4 x [16][NP]
  unsigned* sMask = reinterpret_cast<unsigned*>(sVall + 4 * 16 * NP);   // [4]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  float* sV = sVall + wv * (16 * NP);
  for (int q = tid; q < NP * KP / 4; q += 256) reinterpret_cast<f32x4*>(sM)[q] = reinterpret_cast<const f32x4*>(M)[q];
  for (int q = tid; q < 4 * 16 * NP / 4; q += 256) reinterpret_cast<f32x4*>(sVall)[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < 4) sMask[tid] = 0u;
  __syncthreads();

  const int gw = blockIdx.x * 4 + wv;
  const int b0 = gw * blk_per + (gw < blk_extra ? gw : blk_extra);
  const int nb = blk_per + (gw < blk_extra ? 1 : 0);

  f32x4 P[NT][NTP];
  f32x4 S[NS];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) P[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NS; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_ip = [&](int blk) -> long long {
    return (long long)indptr[(int64_t)blk * 16 + (lane < 17 ? lane : 16)];
  };
  auto base_of = [&](long long ipv) -> long long {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(ipv & 0xffffffffll), 0);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(ipv >> 32), 0);
    return (long long)(((unsigned long long)hi << 32) | lo);
  };
  long long ipv = nb > 0 ? load_ip(b0) : 0;
  long long a = base_of(ipv);
  int rel = (int)(ipv - a);
  int nzb = __builtin_amdgcn_readlane(rel, 16);
  int colv = 0;
  float valv = 0.f;
  if (lane < nzb) { colv = indices[a + lane]; valv = vals[a + lane]; }

  for (int b = 0; b < nb; ++b) {
    const int64_t r0 = (int64_t)(b0 + b) * 16;
    const bool more = b + 1 < nb;
    const long long ipv_n = more ? load_ip(b0 + b + 1) : 0;       // lands during this block

    int rb[17];                                   // row boundaries (scalar)
#pragma unroll
    for (int r = 0; r < 17; ++r) rb[r] = __builtin_amdgcn_readlane(rel, r);

    // ---- (1) scatter into the dense tile + occupancy mask ----
    for (int base_e = 0; base_e < nzb; base_e += 64) {
      const int e = base_e + lane;
      int col = colv;
      float val = valv;
      if (base_e > 0 && e < nzb) { col = indices[a + e]; val = vals[a + e]; }
      if (e < nzb) {
        int row = 0;
#pragma unroll
        for (int r = 1; r < 16; ++r) row += (e >= rb[r]) ? 1 : 0;
        atomicAdd(&sV[row * NP + col], val);                        // duplicates add up
        atomicOr(&sMask[wv], 1u << ((row & 3) * NTP + (col >> 4)));
      }
    }
    // ---- (2) new W rows in A-fragment layout: lane (i, kq) <-> rows 4kq + j ----
    float af[NT][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // this lane group's row 4kq + j spans non-zeros [ea, eb)
      const int ea = kq == 0 ? rb[j] : kq == 1 ? rb[4 + j] : kq == 2 ? rb[8 + j] : rb[12 + j];
      const int eb = kq == 0 ? rb[j + 1] : kq == 1 ? rb[5 + j] : kq == 2 ? rb[9 + j] : rb[13 + j];
      int cmax = rb[j + 1] - rb[j];
      cmax = max(cmax, rb[5 + j] - rb[4 + j]);
      cmax = max(cmax, rb[9 + j] - rb[8 + j]);
      cmax = max(cmax, rb[13 + j] - rb[12 + j]);
      float acc[NT];
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) acc[mt] = 0.f;
      for (int it = 0; it < cmax; ++it) {
        const int e = ea + it;
        const bool on = e < eb;
        int col = __shfl(colv, e & 63, 64);
        float val = __shfl(valv, e & 63, 64);
        if (on && e >= 64) { col = indices[a + e]; val = vals[a + e]; }
        if (!on) { col = 0; val = 0.f; }
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) acc[mt] = fmaf(val, sM[col * KP + 16 * mt + i], acc[mt]);
      }
#pragma unroll
      for (int mt = 0; mt < NT; ++mt) {
        af[mt][j] = acc[mt];
        W[(r0 + 4 * kq + j) * KP + 16 * mt + i] = acc[mt];
      }
    }
    // next block's column/value run: requested now, lands under the MFMAs below
    const long long a_n = base_of(ipv_n);
    const int rel_n = (int)(ipv_n - a_n);
    const int nzb_n = __builtin_amdgcn_readlane(rel_n, 16);
    int colv_n = 0;
    float valv_n = 0.f;
    if (more && lane < nzb_n) { colv_n = indices[a_n + lane]; valv_n = vals[a_n + lane]; }

    // ---- (3) S += W_b^T W_b;  P += W_b^T V_b for the occupied (j, column tile) pairs ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int t = 0;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = mt; nt < NT; ++nt, ++t) S[t] = mfma16(af[mt][j], af[nt][j], S[t]);
    }
    const unsigned occ = __builtin_amdgcn_readfirstlane(sMask[wv]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nt = 0; nt < NTP; ++nt)
        if ((occ >> (j * NTP + nt)) & 1u) {
          const float bf = sV[(4 * kq + j) * NP + 16 * nt + i];
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) P[mt][nt] = mfma16(af[mt][j], bf, P[mt][nt]);
        }
    // ---- (4) restore the all-zero tile and mask ----
    for (int base_e = 0; base_e < nzb; base_e += 64) {
      const int e = base_e + lane;
      int col = colv;
      if (base_e > 0 && e < nzb) col = indices[a + e];
      if (e < nzb) {
        int row = 0;
#pragma unroll
        for (int r = 1; r < 16; ++r) row += (e >= rb[r]) ? 1 : 0;
        sV[row * NP + col] = 0.f;
      }
    }
    if (lane == 0) sMask[wv] = 0u;
    ipv = ipv_n; a = a_n; rel = rel_n; nzb = nzb_n; colv = colv_n; valv = valv_n;
  }

  // ---- sum the 4 waves (serially through LDS: once per launch), wave 0 writes the row-major slab ----
  __syncthreads();
  f32x4* ex = reinterpret_cast<f32x4*>(sm);       // the M image is dead: NT*NTP KiB = its size
  static_assert(NS <= NT * NTP, "S exchange fits into the M image");
  for (int src = 1; src < 4; ++src) {             // P
    if (wv == src) {
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) ex[(mt * NTP + nt) * 64 + lane] = P[mt][nt];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) P[mt][nt] += ex[(mt * NTP + nt) * 64 + lane];
    }
    __syncthreads();
  }
  for (int src = 1; src < 4; ++src) {             // S
    if (wv == src) {
#pragma unroll
      for (int t = 0; t < NS; ++t) ex[t * 64 + lane] = S[t];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int t = 0; t < NS; ++t) S[t] += ex[t * 64 + lane];
    }
    __syncthreads();
  }
  if (wv == 0) {
    const int64_t ldp = (int64_t)NP + KP;
    float* base = slab + (int64_t)blockIdx.x * KP * ldp;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* rowp = base + (int64_t)(16 * mt + 4 * kq + j) * ldp;
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) rowp[16 * nt + i] = P[mt][nt][j];
      }
    int t = 0;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int nt = mt; nt < NT; ++nt, ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) base[(int64_t)(16 * mt + 4 * kq + j) * ldp + NP + 16 * nt + i] = S[t][j];
        if (nt > mt)
          *reinterpret_cast<f32x4*>(base + (int64_t)(16 * nt + i) * ldp + NP + 16 * mt + 4 * kq) = S[t];
      }
  }
}

// Dense image V[m][np] of the CSR rows (V zeroed beforehand): contexts with num_bases > 128 run the dense
// kernels on it.  One thread per row, entries added in storage order (duplicates sum as scipy's toarray()).
__global__ __launch_bounds__(256) void k_csr_densify(const int64_t* __restrict__ indptr,
                                                     const int32_t* __restrict__ indices,
                                                     const float* __restrict__ vals, int64_t m, int np,
                                                     float* __restrict__ V) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= m) return;
  float* row = V + r * np;
  for (int64_t e = indptr[r]; e < indptr[r + 1]; ++e) row[indices[e]] += vals[e];
}
