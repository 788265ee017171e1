// pmf_tiled.h -- general-shape MFMA kernels (any padded m, n; k padded to 16*NT).
//
// These serve every shape the fused one-pass kernel (pmf_fused.h) does not take,
// and the single-hook calls update_w()/update_h() of the reference's plugin API.
//
//   k_rowgemm<NT,EPI>  C[64-row block][KP] = A[rows][K] * B[KP][K]^T  (+ epilogue)
//       EPI_NMF_W : Num = V H^T, Den = W G (G = H H^T), W <- (W*Num)/(Den+1e-9)
//                   pymf/nmf.py:128-132 with (W H) H^T reassociated to W (H H^T)
//       EPI_STORE : C = A B^T            (SNMF V H^T and W1 inv(HH^T), snmf.py:68-70;
//                                          NMFALS V H^T, nmfals.py:88)
//   k_colgemm<NT>      per row-chunk partials of P = W^T V (KP x n) and S = W^T W
//                      (pymf/nmf.py:124-125; snmf.py:79,81; nmfals.py:73,78)
//   k_resid<NT>        sum((V - W H)^2) partials (pymf/nmf.py:110)
#pragma once
#include "pmf_dev.h"
#include "pmf_ipc.h"

enum { EPI_STORE = 0, EPI_NMF_W = 1, EPI_BNMF_W = 2, EPI_RNMF_W = 3,
       EPI_NMF_W_SAVE = 4,     // EPI_NMF_W + store Num = V H^T to C (first iteration of a fixed-H loop)
       EPI_NMF_W_CACHED = 5 }; // Num read back from C: no pass over V (H, hence V H^T, unchanged)

// A [R][64] f32 panel travelling global -> registers -> swizzled LDS, 256 threads.
template <int R>
struct PanelStage {
  static constexpr int CH = (R * 16) / 256 > 0 ? (R * 16) / 256 : 1;
  f32x4 r[CH];
  __device__ __forceinline__ void load(const float* __restrict__ src, int64_t ld, int col0,
                                       int kdim, int tid) {
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int id = tid + 256 * q;
      const int row = id >> 4, c = id & 15;
      const int col = col0 + 4 * c;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < R && col < kdim) v = *reinterpret_cast<const f32x4*>(src + (int64_t)row * ld + col);
      r[q] = v;
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int q = 0; q < CH; ++q) {
      const int id = tid + 256 * q;
      const int row = id >> 4, c = id & 15;
      if (row < R) lds_write4(lds, row, c, r[q]);
    }
  }
};

#ifdef PMF_RG_STAMPS   // diagnostic build only (tools/stamp_rowgemm.hip)
#define PMF_RG_STAMP(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ unsigned long long g_rg_acc[4];   // store, barrier, issue loads, reads+MFMA  (wave 0 of block 0 only)
#else
#define PMF_RG_STAMP(var) do { } while (0)
#endif

// acc[nt] (+)= A_tile[64 x kdim] * B[KP x kdim]^T for this wave's 16 rows.
// sa: 2 x [64][64] floats, sb: 2 x [KP][64] floats.  kdim % 4 == 0.
// Register stages of tile_gemm_nt: panels p (stage 0/1 alternating) on their way global -> LDS.
template <int NT>
struct GemmStages {
  PanelStage<64> pa0, pa1;
  PanelStage<16 * NT> pb0, pb1;
  // request panels 0 and 1 of the product A[64 x kdim] * B[KP x kdim]^T
  __device__ __forceinline__ void prefetch(const float* __restrict__ A, int64_t lda,
                                           const float* __restrict__ B, int64_t ldb, int kdim, int tid) {
    pa0.load(A, lda, 0, kdim, tid);
    pb0.load(B, ldb, 0, kdim, tid);
    if (kdim > 64) {
      pa1.load(A, lda, 64, kdim, tid);
      pb1.load(B, ldb, 64, kdim, tid);
    }
  }
};

// acc[nt] (+)= A_tile[64 x kdim] * B[KP x kdim]^T for this wave's 16 rows; st.prefetch(A, B) has been
// called.  sa: 2 x [64][64] floats, sb: 2 x [KP][64] floats.  kdim % 4 == 0.
// Panels travel global -> registers -> LDS.  TWO register stages: panel p + 2 is requested while
// panel p is multiplied, so a request has two panel times (not one) to come back -- measured with
// in-kernel stamps, one was not enough: the stores to LDS waited ~1 000 cycles per panel.
template <int NT>
__device__ __forceinline__ void tile_gemm_nt(f32x4 (&acc)[NT], const float* __restrict__ A,
                                             int64_t lda, const float* __restrict__ B,
                                             int64_t ldb, int kdim, float* sa, float* sb,
                                             GemmStages<NT>& st) {
  constexpr int KP = 16 * NT;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int npan = (kdim + 63) >> 6;
  auto panel = [&](int p, PanelStage<64>& pa, PanelStage<KP>& pb) {
    float* ca = sa + (p & 1) * (64 * 64);
    float* cb = sb + (p & 1) * (KP * 64);
#ifdef PMF_RG_STAMPS
    unsigned long long t0, t1, t2, t3, t4;
#endif
    PMF_RG_STAMP(t0);
    pa.store(ca, tid);
    pb.store(cb, tid);
    PMF_RG_STAMP(t1);
    __syncthreads();
    PMF_RG_STAMP(t2);
    if (p + 2 < npan) {
      pa.load(A, lda, 64 * (p + 2), kdim, tid);
      pb.load(B, ldb, 64 * (p + 2), kdim, tid);
    }
    PMF_RG_STAMP(t3);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int chunk = 4 * t + kq;
      const f32x4 a4 = lds_read4(ca, 16 * wv + i, chunk);
      f32x4 b4[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b4[nt] = lds_read4(cb, 16 * nt + i, chunk);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(a4[e], b4[nt][e], acc[nt]);
    }
#ifdef PMF_RG_STAMPS
    PMF_RG_STAMP(t4);
    if (blockIdx.x == 100 && tid == 0) { g_rg_acc[0] += t1 - t0; g_rg_acc[1] += t2 - t1; g_rg_acc[2] += t3 - t2; g_rg_acc[3] += t4 - t3; }
#endif
  };
  for (int p = 0; p < npan; p += 2) {
    panel(p, st.pa0, st.pb0);
    if (p + 1 < npan) panel(p + 1, st.pa1, st.pb1);
  }
}

template <int NT>
constexpr size_t rowgemm_smem_bytes() { return (size_t)(2 * 64 * 64 + 2 * 16 * NT * 64) * sizeof(float); }

template <int NT, int EPI>
__global__ __launch_bounds__(256) void k_rowgemm(const float* __restrict__ A, int64_t lda, int kdimA,
                                                 const float* __restrict__ B, int64_t ldb,
                                                 float* __restrict__ W, const float* __restrict__ G,
                                                 float* __restrict__ C, int64_t ldc, float lamb,
                                                 int64_t mvalid, int kvalid, int ntiles, int tpw) {
  constexpr int KP = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sa = smem;
  float* sb = smem + 2 * 64 * 64;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  // A workgroup takes tpw consecutive 64-row tiles: a tile on its own spends ~40 % of its life on the
  // first requests' latency, the epilogue's loads and the workgroup turnover (in-kernel stamps), so the
  // first two panels of tile t + 1 are requested before tile t's epilogue and its old W rows before
  // its Den product.
  const int t_begin = blockIdx.x * tpw;
  int t_end = t_begin + tpw;
  if (t_end > ntiles) t_end = ntiles;
  GemmStages<NT> st;
  if (EPI != EPI_NMF_W_CACHED && t_begin < t_end) st.prefetch(A + (int64_t)t_begin * 64 * lda, lda, B, ldb, kdimA, tid);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int64_t row0 = (int64_t)tile * 64;
    f32x4 num[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) num[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t rbase = row0 + 16 * wv + 4 * kq;   // + reg index j
    if (EPI == EPI_NMF_W_CACHED) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) num[nt][j] = C[(rbase + j) * ldc + 16 * nt + i];
    } else {
      tile_gemm_nt<NT>(num, A + row0 * lda, lda, B, ldb, kdimA, sa, sb, st);
    }
    if (EPI == EPI_NMF_W_SAVE) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) C[(rbase + j) * ldc + 16 * nt + i] = num[nt][j];
    }

    if (EPI != EPI_STORE) {
      f32x4 den[NT];
      float wold[NT][4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        den[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) wold[nt][j] = W[(rbase + j) * KP + 16 * nt + i];   // lands during Den
      }
      __syncthreads();                                     // the LDS panels of the previous product are free
      st.prefetch(W + row0 * KP, KP, G, KP, KP, tid);
      tile_gemm_nt<NT>(den, W + row0 * KP, KP, G, KP, KP, sa, sb, st);
      if (EPI != EPI_NMF_W_CACHED && tile + 1 < t_end)     // next tile's first panels, under the epilogue
        st.prefetch(A + (row0 + 64) * lda, lda, B, ldb, kdimA, tid);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float* p = W + (rbase + j) * KP + 16 * nt + i;
          const float w = wold[nt][j];
          if (EPI == EPI_RNMF_W) {                              // rnmf.py:109-115 (A = S - data, no epsilon)
            const float x = num[nt][j];
            const float r = w * ((fabsf(x) - x) / (2.0f * den[nt][j]));
            *p = ((rbase + j) < mvalid && (16 * nt + i) < kvalid) ? r : 0.f;   // 0/0 on the zero padding
          } else if (EPI == EPI_BNMF_W) {                       // bnmf.py:87-90
            const float w1 = num[nt][j] + (3.0f * lamb) * (w * w);
            const float w2 = ((den[nt][j] + (2.0f * lamb) * (w * w * w)) + lamb * w) + PMF_EPS_DEN;
            *p = w * (w1 / w2);
          } else {
            *p = (w * num[nt][j]) / (den[nt][j] + PMF_EPS_DEN);   // multiply, then divide (nmf.py:131-132)
          }
        }
    } else {
      if (tile + 1 < t_end) st.prefetch(A + (row0 + 64) * lda, lda, B, ldb, kdimA, tid);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) C[(rbase + j) * ldc + 16 * nt + i] = num[nt][j];
    }
    __syncthreads();                                       // LDS panels free for the next tile
  }
}

// N consecutive floats (N = 1, 2, 4, 8; the address is a multiple of 4 N bytes): one or two vector accesses
template <int N>
__device__ __forceinline__ void vec_load(const float* __restrict__ p, float (&v)[N]) {
  if constexpr (N >= 4) {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
      v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
    }
  } else if constexpr (N == 2) {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  } else {
    v[0] = p[0];
  }
}
template <int N>
__device__ __forceinline__ void vec_store(float* __restrict__ p, const float (&v)[N]) {
  if constexpr (N >= 4) {
#pragma unroll
    for (int q = 0; q < N / 4; ++q) *reinterpret_cast<f32x4*>(p + 4 * q) = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
  } else if constexpr (N == 2) {
    *reinterpret_cast<float2*>(p) = float2{v[0], v[1]};
  } else {
    p[0] = v[0];
  }
}

// Plain product C[rows][KP] = A[rows][kdim] B[KP][kdim]^T for long contractions (kdim % 64 == 0): NMFALS' and SNMF's
// V H^T, SNMF's W = V M^T.  The wave keeps 16*RB rows: their A fragments come straight from global memory into two
// register stages (the lane layout of the 16x16x4 MFMA's A operand IS a 16-byte global read per lane), one 64-column
// panel ahead; only B travels through LDS (one [KP][64] panel per barrier, double buffered, shared by the 4 waves).
// The requests are INTERLEAVED with the MFMAs, one b128 request per 4*NT MFMAs: a burst of 20 requests at the top of the
// panel blocks the wave at issue for about as long as the panel's MFMAs take (in-kernel stamps, tools/rowgemm_lab.hip:
// 0.37 ms burst vs 0.31 ms interleaved at 262 144 x 1 024, k = 64; requests only 0.26 ms, MFMAs only 0.27-0.30 ms).
// The B panel for the next barrier is stored at the END of a panel so that its wait sits in straight-line code with an
// exact count of younger requests (at a loop head the compiler falls back to vmcnt(0), draining the prefetch).
// Same k order per accumulator as k_rowgemm: the results are bit-identical.  kdim % 128 == 0 (an even number of
// panels: a tail panel after the loop costs the register allocation of the loop 540 bytes of scratch).
//
// EPI_NMF_W / EPI_BNMF_W / EPI_RNMF_W: Den = W_tile G follows on the same registers (the A stages are dead by then; G
// takes the B buffers' place in LDS, the W fragments come straight from global memory as A did), then the epilogue of
// k_rowgemm, expression for expression.  DENBUF (base blocks beyond 128: W has ldw > KP columns and Den = W G^T was
// formed for all blocks BEFORE the first block of W changes): Den is read from the buffer G points at, [.][ldw] as W.
template <int NT, int RB, int EPI, bool DENBUF = false>
__global__ __launch_bounds__(256, 2) void k_rowgemm_stream(const float* __restrict__ A, int64_t lda, int kdim,
                                                           const float* __restrict__ B, int64_t ldb,
                                                           float* __restrict__ W, const float* __restrict__ G,
                                                           float* __restrict__ C, int64_t ldc, float lamb,
                                                           int64_t mvalid, int kvalid, int ntiles, int64_t ldw) {
  static_assert(EPI == EPI_STORE || EPI == EPI_NMF_W || EPI == EPI_BNMF_W || EPI == EPI_RNMF_W, "k_rowgemm_stream: epilogue");
  constexpr int KP = 16 * NT;
  constexpr int WR = 16 * RB;                          // rows per wave
  constexpr bool DENPROD = EPI != EPI_STORE && !DENBUF;   // Den = W_tile G formed here: G lives in LDS behind the B buffers
  constexpr int GPAN = (KP + 63) / 64;                 // 64-column panels of G
  constexpr int GCH = KP < 64 ? KP / 4 : 16;           // 16-byte chunks per row of a panel
  // up to 64 bases G has a region of its own (staged once); at 128 bases that would be 128 KiB of LDS -- one workgroup per
  // CU --, so G takes the B buffers' place after the product and the launch gives every workgroup ONE group of tiles
  constexpr bool GSEP = NT <= 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sg = GSEP ? smem + 2 * KP * 64 : smem;        // [GPAN][KP][64] (DENPROD only; see rowgemm_stream_smem_bytes)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int npan = kdim >> 6;
  // PERSISTENT workgroups: group g of four wave tiles, g = blockIdx.x, + gridDim.x, ...; the prefetch runs on across the
  // groups (the last panel of a tile requests panel 0 of the wave's next tile), B's panels wrap around, G is staged
  // once.  A tile of a 256-column matrix is four panels: on its own it spent a good part of its life waiting for its
  // first panel and draining its last (1 048 576 x 256, k = 64: 0.50 -> see profiles/r03_experiments.md).
  const int ngroups = (ntiles + 3) >> 2;
  f32x4 acc[RB][NT];
  f32x4 fa0[RB][4], fa1[RB][4];
  constexpr int BCH = KP * 16 / 256;                   // 16-byte pieces of a B panel per thread
  f32x4 pbr[BCH];
  auto load_b = [&](int p) {                           // unconditional: a guard is a branch the waitcnt pass trips over
    const int pp = p < npan ? p : p - npan;            // (wraps: the next tile starts with panel 0 again)
#pragma unroll
    for (int q = 0; q < BCH; ++q) {
      const int id = tid + 256 * q;
      pbr[q] = *reinterpret_cast<const f32x4*>(B + (int64_t)(id >> 4) * ldb + 64 * pp + 4 * (id & 15));
    }
  };
  // The MFMA N index of a lane is free: B row (basis) NT i + nt goes to LDS row 16 nt + i, so column i of tile nt IS basis
  // NT i + nt and the NT tiles of a lane hold NT CONSECUTIVE bases -- the epilogue reads and writes them as vectors
  // (at 128 bases: 512 contiguous bytes per row instead of eight 64-byte pieces).  Same sums, same order: same bits.
  auto store_b = [&](float* cb) {
#pragma unroll
    for (int q = 0; q < BCH; ++q) {
      const int id = tid + 256 * q, brow = id >> 4;
      lds_write4(cb, 16 * (brow % NT) + brow / NT, id & 15, pbr[q]);
    }
  };
  auto stage_g = [&]() {                               // G as [GPAN][KP][64] panel images, rows in the order of B's
    for (int id = tid; id < GPAN * KP * GCH; id += 256) {
      const int pg = id / (KP * GCH), rem = id % (KP * GCH), row = rem / GCH, ch = rem % GCH;
      lds_write4(sg + pg * (KP * 64), 16 * (row % NT) + row / NT, ch, *reinterpret_cast<const f32x4*>(G + row * KP + 64 * pg + 4 * ch));
    }
  };
  if (DENPROD && GSEP) stage_g();
  int tile = blockIdx.x * 4 + wv;                      // tiles of WR rows
  const float* Arow = A + ((int64_t)(tile < ntiles ? tile : 0) * WR + i) * lda + 4 * kq;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int t = 0; t < 4; ++t) fa0[rb][t] = *reinterpret_cast<const f32x4*>(Arow + (int64_t)(16 * rb) * lda + 16 * t);
  load_b(0);
  store_b(smem);
  load_b(1);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const bool act = tile < ntiles;
    const int tile_n = tile + 4 * gridDim.x;           // this wave's tile of the next group (if there is one)
    const bool more = grp + (int)gridDim.x < ngroups;
    const float* Arow_n = A + ((int64_t)(more && tile_n < ntiles ? tile_n : 0) * WR + i) * lda + 4 * kq;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[rb][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto panel = [&](int p, f32x4 (&fa)[RB][4], f32x4 (&fan)[RB][4]) {
      const float* cb = smem + (p & 1) * (KP * 64);
      // next panel of this tile; after the last one, panel 0 of the next tile (or, on the very last, itself once more)
      const float* An = p + 1 < npan ? Arow + 64 * (p + 1) : (more ? Arow_n : Arow + 64 * p);
      __syncthreads();
      // B fragments of step t + 1 are read from LDS under the MFMAs of step t (the scheduling fences below would otherwise
      // put each step's four reads, and their latency, in front of its first MFMA)
      // (up to 64 bases: at 128 the second set of fragments is 32 registers too many beside the epilogues)
      constexpr bool BPRE = NT <= 4;
      f32x4 b4[BPRE ? 2 : 1][NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b4[0][nt] = lds_read4(cb, 16 * nt + i, kq);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (BPRE && t < 3) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) b4[(t + 1) & 1][nt] = lds_read4(cb, 16 * nt + i, 4 * (t + 1) + kq);
        }
        if (!BPRE && t > 0) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) b4[0][nt] = lds_read4(cb, 16 * nt + i, 4 * t + kq);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          fan[rb][t] = *reinterpret_cast<const f32x4*>(An + (int64_t)(16 * rb) * lda + 16 * t);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[rb][nt] = mfma16(fa[rb][t][e], b4[BPRE ? (t & 1) : 0][nt][e], acc[rb][nt]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      store_b(smem + ((p + 1) & 1) * (KP * 64));       // B panel p + 1 (mod npan): its buffer was last read in panel p - 1
      load_b(p + 2);
    };
    for (int p = 0; p < npan; p += 2) {                // no branch INSIDE the pair: LLVM sinks the prefetch into it
      panel(p, fa0, fa1);
      panel(p + 1, fa1, fa0);
    }
    // ---- this tile's epilogue (the next tile's first panel and B panels 0, 1 are on their way) ----
    if (EPI == EPI_STORE) {
      if (act) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) v[nt] = acc[rb][nt][j];
            vec_store<NT>(C + ((int64_t)tile * WR + 16 * rb + 4 * kq + j) * ldc + NT * i, v);
          }
      }
    } else {
      if (DENPROD && !GSEP) {                          // (single group per workgroup: nothing needs the B buffers any more)
        __syncthreads();
        stage_g();
        __syncthreads();
      }
      const float* Wrow = W + ((int64_t)(act ? tile : 0) * WR + i) * ldw + 4 * kq;
      // one block of 16 rows at a time: Den = W_tile G for it (contraction over the KP bases, the W fragments straight from
      // global memory as A was), then its rows' update -- NT accumulators live instead of RB * NT beside the prefetched panel
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        f32x4 den[NT];
        if (DENPROD) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) den[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int pg = 0; pg < GPAN; ++pg)
#pragma unroll
            for (int t = 0; t < GCH / 4; ++t) {
              const f32x4 a4 = *reinterpret_cast<const f32x4*>(Wrow + (int64_t)(16 * rb) * ldw + 64 * pg + 16 * t);
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const f32x4 b4 = lds_read4(sg + pg * (KP * 64), 16 * nt + i, 4 * t + kq);
#pragma unroll
                for (int e = 0; e < 4; ++e) den[nt] = mfma16(a4[e], b4[e], den[nt]);
              }
            }
        }
        if (act) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int64_t row = (int64_t)tile * WR + 16 * rb + 4 * kq + j;
            float* p = W + row * ldw + NT * i;                        // bases NT i .. NT i + NT - 1 of this row
            float wv_[NT], dn_[NT], out[NT];
            vec_load<NT>(p, wv_);
            if (DENBUF) vec_load<NT>(G + row * ldw + NT * i, dn_);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const float w = wv_[nt];
              const float num = acc[rb][nt][j], dn = DENBUF ? dn_[nt] : den[nt][j];
              if (EPI == EPI_RNMF_W) {                              // rnmf.py:109-115 (A = S - data, no epsilon)
                const float r = w * ((fabsf(num) - num) / (2.0f * dn));
                out[nt] = (row < mvalid && (NT * i + nt) < kvalid) ? r : 0.f;   // 0/0 on the zero padding
              } else if (EPI == EPI_BNMF_W) {                       // bnmf.py:87-90
                const float w1 = num + (3.0f * lamb) * (w * w);
                const float w2 = ((dn + (2.0f * lamb) * (w * w * w)) + lamb * w) + PMF_EPS_DEN;
                out[nt] = w * (w1 / w2);
              } else {
                out[nt] = (w * num) / (dn + PMF_EPS_DEN);           // multiply, then divide (nmf.py:131-132)
              }
            }
            vec_store<NT>(p, out);
          }
        }
      }
    }
    tile = tile_n;
    Arow = Arow_n;
  }
}

template <int NT, int EPI, bool DENBUF>
constexpr size_t rowgemm_stream_smem_bytes() {
  return (size_t)(2 * 16 * NT * 64 + ((EPI != EPI_STORE && !DENBUF && NT <= 4) ? ((16 * NT + 63) / 64) * 16 * NT * 64 : 0)) * sizeof(float);
}

// Partials of P = W^T V and S = W^T W over one chunk of rows.
// grid = (nchunks, ceil(np/256)); wave w of a block owns columns [256*by + 64*w, +64).
// slab[chunk][KP][np + KP]: P in columns [0,np), S in [np, np+KP) (written by by == 0).
// The MFMA M / N index of a lane is free, so the operands are fetched with full-width vector
// loads: lane i reads NT consecutive bases (W row = 16*NT floats = one 16-lane row) and 4
// consecutive columns of V; tile e then holds bases {NT*i + e} resp. columns {4*i + e}, and the
// permutation is undone in the slab store.  4 + 4 vector loads feed the 4*NT*(4 + ...) MFMAs of a
// 16-row step (was 16 + 4*NT dword loads).
// WITH_S = false (base blocks beyond 128: S is formed by a pass of its own with W in V's place): no W^T W tiles --
// at NT = 8 they are a third of the MFMAs and 64 of the accumulator registers.
template <int NT, bool WITH_S = true>
__global__ __launch_bounds__(256) void k_colgemm(const float* __restrict__ V, int64_t ldv, int np,
                                                 const float* __restrict__ W, int64_t ldw, int64_t mp,
                                                 int rows_per_chunk, float* __restrict__ slab) {
  constexpr int KP = 16 * NT;
  constexpr int ST = (NT + 3) / 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_chunk;
  int64_t r_end = r_begin + rows_per_chunk;
  if (r_end > mp) r_end = mp;
  const int c0 = blockIdx.y * 256 + 64 * wv;
  const bool pact = (V != nullptr) && c0 < np;   // V == nullptr: S only (CSR path)
  const bool sact = WITH_S && blockIdx.y == 0;

  f32x4 P[NT][4];
  f32x4 S[NT][ST];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) P[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < ST; ++st) S[mt][st] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // The operands of a 16-row step come straight from L2/HBM (full-width vector loads, no LDS): the loads of
  // step r + 16 are issued BEFORE the MFMAs of step r, so a load has one step's MFMAs (512 cycles at NT = 4)
  // plus the other waves' to come back instead of stalling the step it belongs to.
  struct Operands {
    f32x4 w[NT >= 4 ? NT / 4 : 1][4];     // W rows r + 4 kq + j: bases NT i .. NT i + NT - 1
    f32x4 v[4];                           // V rows r + 4 kq + j: columns c0 + 4 i .. + 3
  };
  auto fetch = [&](int64_t r, Operands& o) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = r + 4 * kq + j;   // k index of MFMA step j for this lane group
      const float* wr = W + row * ldw + NT * i;
      if (NT >= 4) {
#pragma unroll
        for (int q = 0; q < NT / 4; ++q) o.w[q][j] = *reinterpret_cast<const f32x4*>(wr + 4 * q);
      } else {
        f32x4 w4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < NT; ++e) w4[e] = wr[e];
        o.w[0][j] = w4;
      }
      o.v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pact) o.v[j] = *reinterpret_cast<const f32x4*>(V + row * ldv + c0 + 4 * i);
    }
  };
  Operands cur, nxt;
  if (r_begin < r_end) fetch(r_begin, cur);
  for (int64_t r = r_begin; r < r_end; r += 16) {
    const bool more = r + 16 < r_end;
    if (more) fetch(r + 16, nxt);
    __builtin_amdgcn_sched_barrier(0);      // keep the requests ahead of this step's MFMAs
    float af[NT][4], bf[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int e = 0; e < NT; ++e) af[e][j] = cur.w[e / 4][j][e % 4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bf[e][j] = cur.v[j][e];
    }
    if (pact) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) P[mt][nt] = mfma16(af[mt][j], bf[nt][j], P[mt][nt]);
    }
    if (sact) {
      // wave w forms the S tiles (mt, nt = w + 4 st); the B operand is the same register file
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int st = 0; st < ST; ++st) {
            float b = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              if (nt == wv + 4 * st) b = af[nt][j];
            S[mt][st] = mfma16(af[mt][j], b, S[mt][st]);
          }
    }
    if (more) cur = nxt;
  }

  // tile (mt, nt), lane (c = i, q = kq), register jj  <->  base NT*(4q + jj) + mt,  column 4c + nt
  const int64_t ldp = (int64_t)np + KP;
  float* base = slab + (int64_t)blockIdx.x * KP * ldp;
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float* rowp = base + (int64_t)(NT * (4 * kq + jj) + mt) * ldp;
      if (pact) {
        f32x4 o;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) o[nt] = P[mt][nt][jj];
        *reinterpret_cast<f32x4*>(rowp + c0 + 4 * i) = o;          // columns 4i .. 4i+3
      }
      if (sact) {
#pragma unroll
        for (int st = 0; st < ST; ++st) {
          const int nt = wv + 4 * st;
          if (nt < NT) rowp[np + NT * i + nt] = S[mt][st][jj];   // column base NT*c + nt
        }
      }
    }
}

// The mirror of k_rowgemm_stream for the partials of P = W^T V (and S = W^T W): the V fragments of a wave's 64 columns go
// straight into two register stages, one stage of SR rows ahead, the requests interleaved with the MFMAs (one b128 per
// 4 NT MFMAs); the W rows of a stage travel through LDS ONCE per workgroup -- k_colgemm's four waves (and the four column
// panels of a 1 024-column V) each fetch them from L2 -- one barrier per stage.  Same slab layout and the same order of
// summation per accumulator as k_colgemm: bit-identical partials.  NT = 4 (SR = 64) and NT = 8 without the S tiles
// (SR = 32: 128 accumulator registers leave room for 2 x 32 of V); rows_per_chunk a multiple of SR.  A chunk with an
// odd number of stages runs one stage more on zeroed W rows (no branch inside the stage pair: LLVM would sink the
// prefetch into it).  262 144 x 1 024, k = 64 in the lab (tools/colgemm_lab.hip): 0.33-0.35 ms against 0.39-0.42.
template <int NT, bool WITH_S>
__global__ __launch_bounds__(256, 2) void k_colgemm_stream(const float* __restrict__ V, int64_t ldv, int np,
                                                           const float* __restrict__ W, int64_t ldw, int64_t mp,
                                                           int rows_per_chunk, float* __restrict__ slab, int64_t ldp,
                                                           int col_off) {
  // ldp: row stride of a slab ([KP][ldp] per chunk), col_off: first slab column of this product (np for the pass that
  // forms S = W^T W as a product of its own, with W in V's place: 64 < num_bases <= 128)
  static_assert(NT == 4 || (NT == 8 && !WITH_S), "k_colgemm_stream: NT = 4, or NT = 8 without the S tiles");
  constexpr int KP = 16 * NT;
  constexpr int ST = (NT + 3) / 4;
  constexpr int SR = NT == 4 ? 64 : 32;                // rows per stage
  constexpr int STEPS = SR / 16;
  constexpr int WLD = KP + 4;                          // padded row of the W stage in LDS (floats)
  constexpr int WCH = SR * (KP / 4) / 256;             // 16-byte pieces of a W stage per thread
  constexpr int AQ = NT / 4;                           // 16-byte pieces of a lane's NT consecutive bases
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_chunk;
  int64_t r_end = r_begin + rows_per_chunk;
  if (r_end > mp) r_end = mp;
  const int nst = (int)((r_end - r_begin) / SR);
  const int c0 = blockIdx.y * 256 + 64 * wv;
  const bool pact = c0 < np;
  const bool sact = WITH_S && blockIdx.y == 0;
  const int c0l = pact ? c0 : np - 64;
  f32x4 P[NT][4];
  f32x4 S[NT][ST];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) P[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < ST; ++st) S[mt][st] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* Vl = V + (r_begin + 4 * kq) * ldv + c0l + 4 * i;
  f32x4 va0[STEPS][4], va1[STEPS][4];                  // [16-row step][j]: V[r + 16 t + 4 kq + j][c0 + 4 i ..]
  f32x4 pw[WCH];
  auto load_w = [&](int s) {                           // (a stage beyond the chunk re-requests the last one: straight-line code)
    const int ss = s < nst ? s : nst - 1;
#pragma unroll
    for (int q = 0; q < WCH; ++q) {
      const int id = tid + 256 * q;
      pw[q] = *reinterpret_cast<const f32x4*>(W + (r_begin + (int64_t)ss * SR + id / (KP / 4)) * ldw + 4 * (id % (KP / 4)));
    }
  };
  auto store_w = [&](float* buf, int s) {
    const bool live = s < nst;                         // ... and multiplies by zeros
#pragma unroll
    for (int q = 0; q < WCH; ++q) {
      const int id = tid + 256 * q;
      const f32x4 v = live ? pw[q] : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(buf + (id / (KP / 4)) * WLD + 4 * (id % (KP / 4))) = v;
    }
  };
  struct AFrag { f32x4 q[AQ]; };                       // bases NT i .. NT i + NT - 1 of one W row
  auto read_a = [&](const float* wb, int row) {
    AFrag a;
#pragma unroll
    for (int u = 0; u < AQ; ++u) a.q[u] = *reinterpret_cast<const f32x4*>(wb + row * WLD + NT * i + 4 * u);
    return a;
  };
  auto stage = [&](int s, f32x4 (&va)[STEPS][4], f32x4 (&van)[STEPS][4]) {
    const float* wb = smem + (s & 1) * (SR * WLD);
    const int sn = s + 1 < nst ? s + 1 : nst - 1;
    const float* Vn = Vl + (int64_t)sn * SR * ldv;
    __syncthreads();
    AFrag a = read_a(wb, 4 * kq);
#pragma unroll
    for (int t = 0; t < STEPS; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        AFrag an = a;
        if (4 * t + j < 4 * STEPS - 1) {                 // the next step's W fragment: its LDS round trip runs under these MFMAs
          const int tn = (4 * t + j + 1) >> 2, jn = (4 * t + j + 1) & 3;
          an = read_a(wb, 16 * tn + 4 * kq + jn);
        }
        van[t][j] = *reinterpret_cast<const f32x4*>(Vn + (int64_t)(16 * t + j) * ldv);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) P[mt][nt] = mfma16(a.q[mt / 4][mt % 4], va[t][j][nt], P[mt][nt]);
        if (sact) {
          // wave w forms the S tiles (mt, nt = w + 4 st); the B operand is the same register file
#pragma unroll
          for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int st = 0; st < ST; ++st) {
              float b = 0.f;
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                if (nt == wv + 4 * st) b = a.q[nt / 4][nt % 4];
              S[mt][st] = mfma16(a.q[mt / 4][mt % 4], b, S[mt][st]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        a = an;
      }
    store_w(smem + ((s + 1) & 1) * (SR * WLD), s + 1);  // W stage s + 1: its buffer was last read in stage s - 1
    load_w(s + 2);
  };
#pragma unroll
  for (int t = 0; t < STEPS; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) va0[t][j] = *reinterpret_cast<const f32x4*>(Vl + (int64_t)(16 * t + j) * ldv);
  load_w(0);
  store_w(smem, 0);
  load_w(1);
  for (int s = 0; s < nst; s += 2) {
    stage(s, va0, va1);
    stage(s + 1, va1, va0);
  }
  // tile (mt, nt), lane (c = i, q = kq), register jj  <->  base NT*(4q + jj) + mt,  column 4c + nt  (as k_colgemm)
  float* base = slab + (int64_t)blockIdx.x * KP * ldp + col_off;
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float* rowp = base + (int64_t)(NT * (4 * kq + jj) + mt) * ldp;
      if (pact) {
        f32x4 o;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) o[nt] = P[mt][nt][jj];
        *reinterpret_cast<f32x4*>(rowp + c0 + 4 * i) = o;
      }
      if (sact) {
#pragma unroll
        for (int st = 0; st < ST; ++st) {
          const int nt = wv + 4 * st;
          if (nt < NT) rowp[np - col_off + NT * i + nt] = S[mt][st][jj];
        }
      }
    }
}

// out[e] = sum over slabs of slab[c][e], fixed order, float64 accumulation.
// One block (1024 threads) = 256 consecutive elements as 64 float4; wave w sums slabs
// w, w+16, ... (coalesced 1-KiB reads), the 16 partials are combined in wave order.
// Gd (may be null; NMFALS on one rank): the S = W^T W part of the sums also goes out as the float64 Hessian of the column
// QPs, [KP][KP] with the identity on the padding (nmfals.py:78) -- k_hessian_from_ps's job without a launch of its own.
__global__ __launch_bounds__(1024) void k_reduce_slabs(const float* __restrict__ slab, int nslabs,
                                                       int64_t E, float* __restrict__ out,
                                                       double* __restrict__ Gd = nullptr, int np = 0, int KP = 0, int k = 0) {
  __shared__ double part[16][64][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t e4 = (int64_t)blockIdx.x * 64 + lane;      // float4 index
  const int64_t E4 = E >> 2;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (e4 < E4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(slab) + e4;
#pragma unroll 4
    for (int c = wv; c < nslabs; c += 16) {
      const f32x4 v = p[(int64_t)c * E4];
      s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
    }
  }
  part[wv][lane][0] = s0; part[wv][lane][1] = s1; part[wv][lane][2] = s2; part[wv][lane][3] = s3;
  __syncthreads();
  if (wv < 4 && e4 < E4) {        // wave q combines component q of the 64 float4
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w][lane][wv];
    out[4 * e4 + wv] = (float)t;
    if (Gd != nullptr) {
      const int64_t e = 4 * e4 + wv, ldp = (int64_t)np + KP;
      const int r = (int)(e / ldp), cc = (int)(e % ldp) - np;
      if (cc >= 0) Gd[(int64_t)r * KP + cc] = (r < k && cc < k) ? (double)(float)t : (r == cc ? 1.0 : 0.0);
    }
  }
}

// Fused-kernel slabs are tile-major (pmf_fused.h): block b sums tile b of every slab (float64,
// fixed order, wave w takes slabs w, w+16, ...) and scatters it into the row-major (P | S)
// buffer; S tiles above the diagonal are mirrored.
// pr.nranks > 1 (round 5, the folded exchange): the reduced tile is this rank's PARTIAL of the cross-rank sum -- instead of
// `out` it goes straight into slot [seq & 1][me] of every rank's receive area (same row-major (P | S) indices), flag `tile`
// is raised there, and the consumer (k_nmf_h_gram's prologue) adds the N partials in rank order: no launch for the exchange.
__global__ __launch_bounds__(1024) void k_reduce_slabs_tiles(const float* __restrict__ slab, int nslabs,
                                                             int NT, int NTP, int np,
                                                             float* __restrict__ out,
                                                             const int* __restrict__ stop, IpcPeers pr, unsigned seq) {
  __shared__ double part[16][64][4];
  if (stop != nullptr && *stop != 0) return;       // (rank-consistent: every rank skips the push AND the wait of this exchange)
  const bool push = pr.nranks > 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int KP = 16 * NT;
  const int NTU = NT * NTP + NT * (NT + 1) / 2;
  const int tile = blockIdx.x;
  const f32x4* p = reinterpret_cast<const f32x4*>(slab) + (size_t)tile * 64 + lane;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  // (round 6: sixteen loads in flight per wave -- 256 slabs are one batch, not four dependent round trips to memory the
  //  one-pass kernel's workgroups all over the chip have just written; the order of the additions is what it was)
  int c = wv;
  for (; c + 16 * 15 < nslabs; c += 256) {
    f32x4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(size_t)(c + 16 * u) * NTU * 64];
#pragma unroll
    for (int u = 0; u < 16; ++u) { s0 += (double)v[u][0]; s1 += (double)v[u][1]; s2 += (double)v[u][2]; s3 += (double)v[u][3]; }
  }
#pragma unroll 4
  for (; c < nslabs; c += 16) {
    const f32x4 v = p[(size_t)c * NTU * 64];
    s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
  }
  part[wv][lane][0] = s0; part[wv][lane][1] = s1; part[wv][lane][2] = s2; part[wv][lane][3] = s3;
  __syncthreads();
  if (wv < 4) {                      // wave r combines register r of the tile
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w][lane][wv];
    const float v = (float)t;
    const int i = lane & 15, kq = lane >> 4, r = wv;
    const int64_t ldp = (int64_t)np + KP;
    if (tile < NT * NTP) {
      // P tile (mt, nt = 4p + e) of the fused kernel holds columns {64p + 4c + e} (lane c) and, in
      // tile row m, basis NT m + mt (the NT tiles of a lane are NT consecutive bases)
      const int mt = tile / NTP, nt = tile % NTP;
      const int64_t e0 = (int64_t)(NT * (4 * kq + r) + mt) * ldp + 64 * (nt >> 2) + 4 * i + (nt & 3);
      if (push) ipc_push_f32(pr, seq, e0, v); else out[e0] = v;
    } else {
      int sidx = tile - NT * NTP, mt = 0;
      while (sidx >= NT - mt) { sidx -= NT - mt; ++mt; }
      const int nt = mt + sidx;
      const int row = NT * (4 * kq + r) + mt, col = NT * i + nt;
      const int64_t e0 = (int64_t)row * ldp + np + col, e1 = (int64_t)col * ldp + np + row;
      if (push) { ipc_push_f32(pr, seq, e0, v); if (nt > mt) ipc_push_f32(pr, seq, e1, v); }
      else { out[e0] = v; if (nt > mt) out[e1] = v; }
    }
  }
  if (push) ipc_raise(pr, seq, tile);
}

// Residual pass over 64 rows per block: R = V - W H with W H on MFMA.
//   RNMF = false: part[block] = sum(R^2) in float64                  (pymf/nmf.py:110)
//   RNMF = true : additionally S = soft_threshold(R, lamb) and D = S - V is stored
//                 (rnmf.py:96-98; D is what both RNMF contractions use, rnmf.py:102,111).
// The W fragments of the wave's 16 rows stay in registers; the H panel [KP][64] is staged in LDS
// (double buffered, one barrier per panel; rows padded to 80 floats so the 4 k-rows of a
// fragment read fall on distinct banks); the V values of a panel are requested before its MFMAs.
template <int NT>
constexpr size_t resid_smem_bytes() { return (size_t)2 * 16 * NT * 80 * sizeof(float); }

template <int NT, bool RNMF>
__global__ __launch_bounds__(256) void k_resid(const float* __restrict__ V, int64_t ldv, int np,
                                               const float* __restrict__ W,
                                               const float* __restrict__ H, int64_t ldh, float lamb,
                                               float* __restrict__ D, double* __restrict__ part) {
  constexpr int KP = 16 * NT, HP = 80;
  extern __shared__ __attribute__((aligned(16))) float hsm[];     // 2 x [KP][HP]
  __shared__ double wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 64 + 16 * wv;
  float af[4 * NT];                                   // A[i = row][k = 4s + kq]
#pragma unroll
  for (int s = 0; s < 4 * NT; ++s) af[s] = W[(row0 + i) * KP + 4 * s + kq];
  f32x4 hreg[NT];                                     // this thread's share of the next H panel
  auto hload = [&](int cp) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int id = tid + 256 * q, kk = id >> 4, c4 = id & 15;
      hreg[q] = *reinterpret_cast<const f32x4*>(H + (int64_t)kk * ldh + cp + 4 * c4);
    }
  };
  auto hstore = [&](float* dst) {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int id = tid + 256 * q, kk = id >> 4, c4 = id & 15;
      *reinterpret_cast<f32x4*>(dst + kk * HP + 4 * c4) = hreg[q];
    }
  };
  double tot = 0.0;
  hload(0);
  int buf = 0;
  for (int cp = 0; cp < np; cp += 64, buf ^= 1) {
    float* hp = hsm + buf * (KP * HP);
    hstore(hp);
    __syncthreads();
    if (cp + 64 < np) hload(cp + 64);
    // tile e holds columns {4i + e} (the MFMA N index of a lane is free): V, the H fragments and
    // the D store are all 16-byte accesses
    f32x4 vv[4];                                       // vv[j] = V[row 4kq + j][cp + 4i .. +3]
#pragma unroll
    for (int j = 0; j < 4; ++j)
      vv[j] = *reinterpret_cast<const f32x4*>(V + (row0 + 4 * kq + j) * ldv + cp + 4 * i);
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4 * NT; ++s) {
      const f32x4 hq = *reinterpret_cast<const f32x4*>(hp + (4 * s + kq) * HP + 4 * i);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = mfma16(af[s], hq[e], acc[e]);
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 dq;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = vv[j][e];
        const float r = v - acc[e][j];
        ss += r * r;
        float sv = 0.f;                             // soft thresholding, rnmf.py:75-79
        if (r > lamb) sv = r - lamb;
        else if (r < -lamb) sv = r + lamb;
        dq[e] = sv - v;
      }
      if (RNMF) *reinterpret_cast<f32x4*>(D + (row0 + 4 * kq + j) * ldv + cp + 4 * i) = dq;
    }
    tot += (double)ss;
  }
  tot = wave_sum_f64(tot);
  if (lane == 0) wsum[wv] = tot;
  __syncthreads();
  if (tid == 0) part[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// The same pass with H RESIDENT in LDS ([KP][np + 16] floats: up to 150 KiB) and persistent workgroups: k_resid stages
// the H panels per 64-row tile (as many bytes of H from L2 as of V from HBM at cfg4 size, one barrier per 64 MFMAs); here
// H is staged once per workgroup, the waves run free over their tiles, the V values of the next panel are requested before
// the current panel's MFMAs.  Same arithmetic per element; one float64 partial per workgroup (its waves' sums over their
// tiles, added in wave order): part[blockIdx.x] -- the grid is a fixed 512, not a property of the device.
template <int NT>
constexpr size_t resid_res_smem_bytes(int np) { return (size_t)16 * NT * (np + 16) * sizeof(float); }

template <int NT, bool RNMF>
__global__ __launch_bounds__(256, 2) void k_resid_res(const float* __restrict__ V, int64_t ldv, int np,
                                                      const float* __restrict__ W,
                                                      const float* __restrict__ H, int64_t ldh, float lamb,
                                                      float* __restrict__ D, double* __restrict__ part, int ntiles) {
  constexpr int KP = 16 * NT;
  extern __shared__ __attribute__((aligned(16))) float hsm[];     // [KP][np + 16]
  const int HLD = np + 16;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, kq = lane >> 4;
  for (int id = tid; id < KP * (np / 4); id += 256) {
    const int kk = id / (np / 4), c4 = id % (np / 4);
    *reinterpret_cast<f32x4*>(hsm + kk * HLD + 4 * c4) = *reinterpret_cast<const f32x4*>(H + (int64_t)kk * ldh + 4 * c4);
  }
  __syncthreads();
  __shared__ double wsum[4];
  double wtot = 0.0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * 64 + 16 * wv;
    float af[4 * NT];                                   // A[i = row][k = 4s + kq]
#pragma unroll
    for (int s = 0; s < 4 * NT; ++s) af[s] = W[(row0 + i) * KP + 4 * s + kq];
    const float* Vr = V + (row0 + 4 * kq) * ldv + 4 * i;
    float* Dr = RNMF ? D + (row0 + 4 * kq) * ldv + 4 * i : nullptr;
    f32x4 vn[4];                                        // vn[j] = V[row 4kq + j][cp + 4i .. +3] of the NEXT panel
#pragma unroll
    for (int j = 0; j < 4; ++j) vn[j] = *reinterpret_cast<const f32x4*>(Vr + (int64_t)j * ldv);
    double tot = 0.0;
    for (int cp = 0; cp < np; cp += 64) {
      f32x4 vv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) vv[j] = vn[j];
      const int cn = cp + 64 < np ? cp + 64 : cp;       // (the last panel re-requests itself)
#pragma unroll
      for (int j = 0; j < 4; ++j) vn[j] = *reinterpret_cast<const f32x4*>(Vr + (int64_t)j * ldv + cn);
      f32x4 acc[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* hp = hsm + cp + 4 * i;
#pragma unroll
      for (int s = 0; s < 4 * NT; ++s) {
        const f32x4 hq = *reinterpret_cast<const f32x4*>(hp + (4 * s + kq) * HLD);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = mfma16(af[s], hq[e], acc[e]);
      }
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 dq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = vv[j][e];
          const float r = v - acc[e][j];
          ss += r * r;
          float sv = 0.f;                             // soft thresholding, rnmf.py:75-79
          if (r > lamb) sv = r - lamb;
          else if (r < -lamb) sv = r + lamb;
          dq[e] = sv - v;
        }
        if (RNMF) *reinterpret_cast<f32x4*>(Dr + (int64_t)j * ldv + cp) = dq;
      }
      tot += (double)ss;
    }
    wtot += wave_sum_f64(tot);
  }
  if (lane == 0) wsum[wv] = wtot;
  __syncthreads();
  if (tid == 0) part[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ---- num_bases > 128 (NMF): the bases are handled in blocks of 128 by the NT = 8 kernels above ----
// out[r * out_ld + c] = sum over slabs of slab[s][r][c]  (r < rows, c < ncols; slab rows have src_ld floats)
// One block (1024 threads) = 64 float4 of the [rows][ncols] result; wave w sums slabs w, w + 16, ... in float64 (coalesced
// reads), the 16 partials are combined in wave order (as k_reduce_slabs).  ncols % 4 == 0.
// (One thread per element walking the 1 024 slabs one dword at a time took 364 us for 201 MB: 0.55 TB/s.)
// OutT = double with accumulate != 0: out += the sum (the float64 image of a streamed pass, tile after tile).
template <typename OutT>
__global__ __launch_bounds__(1024) void k_reduce_slabs_block(const float* __restrict__ slab, int nslabs, int rows,
                                                             int src_ld, int ncols, OutT* __restrict__ out,
                                                             int64_t out_ld, int accumulate) {
  __shared__ double part[16][64][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int nc4 = ncols >> 2;
  const int64_t e4 = (int64_t)blockIdx.x * 64 + lane;
  const bool ok = e4 < (int64_t)rows * nc4;
  const int r = ok ? (int)(e4 / nc4) : 0, c = ok ? 4 * (int)(e4 % nc4) : 0;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (ok) {
    const float* p = slab + (int64_t)r * src_ld + c;
    const int64_t slab_stride = (int64_t)rows * src_ld;
#pragma unroll 4
    for (int sl = wv; sl < nslabs; sl += 16) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + sl * slab_stride);
      s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2]; s3 += (double)v[3];
    }
  }
  part[wv][lane][0] = s0; part[wv][lane][1] = s1; part[wv][lane][2] = s2; part[wv][lane][3] = s3;
  __syncthreads();
  if (wv == 0 && ok) {
    OutT* o = out + (int64_t)r * out_ld + c;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      double t = part[0][lane][q];
#pragma unroll
      for (int w = 1; w < 16; ++w) t += part[w][lane][q];
      o[q] = accumulate ? (OutT)((double)o[q] + t) : (OutT)t;
    }
  }
}

// W <- (W * Num) / (Den + 1e-9), elementwise over [rows][ld] (pymf/nmf.py:128-132); the zero padding stays 0
// bnmf == 1: the BNMF rule W *= (Num + 3 l W^2) / (Den + 2 l W^3 + l W + 1e-9) (bnmf.py:87-90)
// bnmf == 2: the RNMF rule W *= (|Num| - Num) / (2 Den) with Num = (S - data) H^T, no epsilon (rnmf.py:109-115);
//            rows >= mvalid and columns >= kvalid of the [.][KP] buffers are padding (0/0 there): kept 0
__global__ __launch_bounds__(256) void k_nmf_w_elem(float* __restrict__ W, const float* __restrict__ Num,
                                                    const float* __restrict__ Den, int64_t count, int bnmf,
                                                    float lamb, int KP = 0, int64_t mvalid = 0, int kvalid = 0) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < count; q += (int64_t)gridDim.x * 256) {   // (grid-stride: > 2^32 elements)
    const float w = W[q];
    if (bnmf == 2) {
      const float x = Num[q];
      const float r = w * ((fabsf(x) - x) / (2.0f * Den[q]));
      W[q] = (q / KP < mvalid && (int)(q % KP) < kvalid) ? r : 0.f;
    } else if (bnmf) {
      const float w1 = Num[q] + (3.0f * lamb) * (w * w);
      const float w2 = ((Den[q] + (2.0f * lamb) * (w * w * w)) + lamb * w) + PMF_EPS_DEN;
      W[q] = w * (w1 / w2);
    } else {
      W[q] = (w * Num[q]) / (Den[q] + PMF_EPS_DEN);
    }
  }
}

// Direct residual for num_bases > 128: part[block] = sum((V - W H)^2) over a 64 x 64 tile, plain float32
// FMAs with LDS staging (a fallback for nearly exact fits, where the trace identity cancels -- not a hot path).
// RNMF = true: additionally D = soft_threshold(V - W H, lamb) - V is stored (as k_resid<NT, true>).
template <bool RNMF>
__global__ __launch_bounds__(256) void k_resid_bigk(const float* __restrict__ V, int64_t ldv,
                                                    const float* __restrict__ W, int KP,
                                                    const float* __restrict__ H, int64_t ldh,
                                                    double* __restrict__ part, float lamb = 0.f,
                                                    float* __restrict__ D = nullptr) {
  __shared__ float Ws[64][17];
  __shared__ float Hs[16][64];
  __shared__ double wsum[4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t r0 = (int64_t)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  for (int k0 = 0; k0 < KP; k0 += 16) {
    for (int q = tid; q < 64 * 16; q += 256) Ws[q >> 4][q & 15] = W[(r0 + (q >> 4)) * KP + k0 + (q & 15)];
    for (int q = tid; q < 16 * 64; q += 256) Hs[q >> 6][q & 63] = H[(int64_t)(k0 + (q >> 6)) * ldh + c0 + (q & 63)];
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float wv[4], hv[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) wv[a] = Ws[ty + 16 * a][kk];
#pragma unroll
      for (int b = 0; b < 4; ++b) hv[b] = Hs[kk][tx + 16 * b];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(wv[a], hv[b], acc[a][b]);
    }
    __syncthreads();
  }
  double s = 0.0;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const float v = V[(r0 + ty + 16 * a) * ldv + c0 + tx + 16 * b];
      const float r = v - acc[a][b];
      s += (double)r * (double)r;
      if (RNMF) {                                  // soft thresholding, rnmf.py:75-79
        float sv = 0.f;
        if (r > lamb) sv = r - lamb;
        else if (r < -lamb) sv = r + lamb;
        D[(r0 + ty + 16 * a) * ldv + c0 + tx + 16 * b] = sv - v;
      }
    }
  s = wave_sum_f64(s);
  if ((tid & 63) == 0) wsum[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
