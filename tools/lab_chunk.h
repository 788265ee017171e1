// lab_chunk.h -- LAB kernels (round 5; measured, bit-identical to the production kernels, NOT faster: profiles/r05_experiments.md --
// kept as the A/B record, not part of libpymf_hip): the two long products of a two-pass iteration at 64 bases in the one-pass
// kernel's style.
//
//   k_colgemm_chunk   partials of P = W^T V (k x n) and S = W^T W       pymf/nmfals.py:73,78 (nmf.py:124-125, snmf.py:79,81)
//   k_rowgemm_chunk   C = V H^T (m x k)                                 pymf/nmfals.py:88    (snmf.py:68)
//
// k_rowgemm_stream / k_colgemm_stream (pmf_tiled.h) run two waves per SIMD around barriers, their V fragments straight from
// global memory in 64-byte pieces per row: 0.77 of the fp32 MFMA peak at 262 144 x 1 024, k = 64, with 9 % / 18 % more HBM
// traffic than the data (profiles/r04_cfg3_kernel_stats.csv).  The one-pass kernel (pmf_fused.h) holds 0.9 on the same chip
// with ONE wave per SIMD, no barrier in its loop, V tiles by LDS-DMA in 1-KiB requests and a 16-byte LDS read per 4 MFMAs.
// These two kernels are its phase B and its phase A for matrices too wide for it (H does not fit LDS beyond 256 columns at
// 64 bases), cut into CHUNKS of 256 columns:
//
//  * k_colgemm_chunk: wave w of a workgroup owns column chunk w of a 1 024-column group for ALL 16-row blocks of the
//    workgroup's row range -- P of a chunk is 64 tiles = 256 accumulator registers, so the four waves together hold
//    P[64][1 024] for the whole launch: no cross-wave sum, no barrier, one slab store at the end.  Per block the wave
//    takes the W rows (16 x 64) and its 16 x 256 piece of V by LDS-DMA into double-buffered private images (a block
//    ahead; V one request per 16 MFMAs), reads the W rows in the MFMA C layout (four ds_read_b128), then runs 16 steps of
//    one ds_read_b128 + 16 MFMAs.  The ten S tiles are split 3 / 3 / 2 / 2 over the waves.
//  * k_rowgemm_chunk: H's chunk c (64 x 256: 64 KiB) is shared in LDS by the four waves; every wave keeps the accumulators
//    of GB = 8 of its 16-row blocks in registers and sweeps them against chunk 0, then 1, 2, 3 (two workgroup barriers per
//    chunk switch, i.e. per 65 k MFMA cycles); V pieces as above.  Same k order per accumulator as k_rowgemm(_stream).
//
// Column i of tile nt is basis NT i + nt throughout (a lane's NT tiles are NT consecutive bases), as everywhere else.
#pragma once
#include "../pymf_amd/csrc/pmf_dev.h"

#ifndef PMF_GLDS16
#define PMF_GLDS16(gsrc, ldst)                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc), \
                                   (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)
#endif

namespace pmf_chunk {
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// the V tile's swizzle (pmf_fused.h): conflict-free for the row-wise reads of phase B and the column-wise reads of phase A
__device__ __forceinline__ int vx(int row) { return (row ^ ((row & 4) << 1)) & 15; }
__device__ __forceinline__ int voff(int row, int chunk) { return row * 64 + ((chunk ^ vx(row)) << 2); }
}  // namespace pmf_chunk

constexpr size_t colgemm_chunk_smem_bytes() { return (size_t)4 * 2 * (4 * 1024 + 1024) * sizeof(float); }   // 4 waves x 2 buffers x ([4][16][64] of V + [16][64] of W): 160 KiB

// V [mp][ldv] (np columns, np % 1024 == 0: blockIdx.y = 1 024-column group), W [mp][ldw] (64 bases), rows_per_wg % 16 == 0.
// slab [gridDim.x][64][ldp]: P in columns [0, np), S in [np, np + 64) (written by blockIdx.y == 0), as k_colgemm(_stream).
__global__ __launch_bounds__(256, 1) void k_colgemm_chunk(const float* __restrict__ V, int64_t ldv, int np,
                                                          const float* __restrict__ W, int64_t ldw, int64_t mp,
                                                          int rows_per_wg, float* __restrict__ slab, int64_t ldp
#ifdef PMF_CHUNK_STAMPS                               // diagnostic build (tools/gemm_ab.hip): cycles and wall clock of the block loop
                                                          , unsigned long long* __restrict__ dbg
#endif
                                                          ) {
  constexpr int NT = 4, KP = 64;
  using namespace pmf_chunk;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  float* sV = smem + wv * (2 * 4096);
  float* sW = smem + 4 * (2 * 4096) + wv * (2 * 1024);
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_wg;
  int64_t r_end = r_begin + rows_per_wg;
  if (r_end > mp) r_end = mp;
  const int nb = r_end > r_begin ? (int)((r_end - r_begin) / 16) : 0;
  const int col0 = 1024 * blockIdx.y + 256 * wv;
  const bool sact = blockIdx.y == 0;

  // LDS-DMA geometry: one request = 4 rows x 256 B of one 64-column panel; lane L fills physical chunk (L & 15) of row 4q + (L >> 4)
  unsigned vo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * q + (lane >> 4);
    vo[q] = (unsigned)(row * (int)ldv * 4 + 16 * ((lane & 15) ^ vx(row)));
  }
  const char* Vb = reinterpret_cast<const char*>(V) + ((size_t)r_begin * ldv + col0) * 4;
  auto issue_v = [&](int blk, int s, int buf) {       // request s (panel s / 4, row group s % 4) of block blk
#ifdef PMF_CHUNK_ABLATE_DMA                           // timing-only diagnostic build (tools/gemm_ab.hip): outputs are wrong
    if (blk != 0) return;
#endif
    PMF_GLDS16(Vb + (size_t)blk * (16 * ldv * 4) + (s >> 2) * 256 + vo[s & 3], sV + buf * 4096 + (s >> 2) * 1024 + (s & 3) * 256);
  };
  // The W rows of a block travel by LDS-DMA as well (a [16][64] image, V's swizzle): a register-destination load in the loop
  // makes the compiler put its own s_waitcnt vmcnt(0) in front of the first MFMA that reads it -- across the loop's back edge it
  // cannot count the requests in between -- and that drains the V prefetch once per block.
  unsigned wo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * q + (lane >> 4);
    wo[q] = (unsigned)(row * (int)ldw * 4 + 16 * ((lane & 15) ^ vx(row)));
  }
  const char* Wb = reinterpret_cast<const char*>(W) + (size_t)r_begin * ldw * 4;
  auto issue_w = [&](int blk, int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) PMF_GLDS16(Wb + (size_t)blk * (16 * ldw * 4) + wo[q], sW + buf * 1024 + q * 256);
  };

  f32x4 P[NT][16];
  f32x4 S[10];                                        // the S tiles on / above the diagonal, over THIS wave's share of the rows (below)
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) P[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 10; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nb > 0) {
    issue_w(0, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) issue_v(0, s, 0);
  }
  // S = W^T W is a sum over rows: wave w takes the rows {4 kq + w} of every block (MFMA step j = w of the four), all ten tiles
  // on / above the diagonal -- 10 MFMAs per block and wave, the same code in every wave (the wave's rows come from the W image
  // by a read of their own); the four partial sums are added through LDS at the end.
  auto block = [&](int b) {
    const float* img = sV + (b & 1) * 4096;
    const float* wim = sW + (b & 1) * 1024;
    const int nbuf = (b + 1) & 1;
    const int bn = b + 1 < nb ? b + 1 : b;            // (the last block re-requests itself: straight-line code)
    // Younger than this block's panel 0: its panels 1-3 (12 requests).  The W rows of this block are older.
    wait_vm<12>();
    issue_w(bn, nbuf);
    f32x4 a[4];                                       // [j]: bases NT i .. NT i + 3 of row 4 kq + j (the MFMA C layout's rows)
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f32x4*>(wim + voff(4 * kq + j, i));
    const f32x4 as = *reinterpret_cast<const f32x4*>(wim + voff(4 * kq + wv, i));
    f32x4 bf[2];
    bf[0] = *reinterpret_cast<const f32x4*>(img + voff(4 * kq, i));
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int p = s >> 2, j = s & 3;
      if (s + 1 < 16) {
        // panel p + 1 of this block has landed: younger are the panels behind it, the 4 W loads and the requests of the steps so far
        if (((s + 1) & 3) == 0) wait_vm<15>();   // (8 + 4 + 3, 4 + 4 + 7, 0 + 4 + 11)
        bf[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(img + ((s + 1) >> 2) * 1024 + voff(4 * kq + ((s + 1) & 3), i));
      }
      const f32x4 v = bf[s & 1];
#ifdef PMF_CHUNK_ABLATE_MFMA                          // timing-only diagnostic build: one MFMA per step instead of sixteen
      P[0][4 * p] = mfma16(a[j][0], v[0] + v[1] + v[2] + v[3], P[0][4 * p]);
#else
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) P[mt][4 * p + nt] = mfma16(a[j][mt], v[nt], P[mt][4 * p + nt]);
      if (sact && s == 0) {
        int t = 0;
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int nt = mt; nt < NT; ++nt) { S[t] = mfma16(as[mt], as[nt], S[t]); ++t; }
      }
#endif
      issue_v(bn, s, nbuf);
      if (s + 1 < 16) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
#ifdef PMF_CHUNK_STAMPS
  unsigned long long tc0, tc1;
  const unsigned long long tw0 = wall_clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc0)::"memory");
#endif
  int b = 0;
  for (; b + 1 < nb; b += 2) {                        // (pairs: the buffer index is a constant in each half)
    block(b);
    block(b + 1);
  }
  if (b < nb) block(b);
  wait_vm<0>();
#ifdef PMF_CHUNK_STAMPS
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tc1)::"memory");
  if (lane == 0 && blockIdx.y == 0) { dbg[(blockIdx.x * 4 + wv) * 2] = tc1 - tc0; dbg[(blockIdx.x * 4 + wv) * 2 + 1] = wall_clock64() - tw0; }
#endif

  // tile (mt, 4 p + e), lane (i, kq), register jj  <->  basis NT (4 kq + jj) + mt,  column col0 + 64 p + 4 i + e
  float* base = slab + (int64_t)blockIdx.x * KP * ldp;
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      float* rowp = base + (int64_t)(NT * (4 * kq + jj) + mt) * ldp + col0 + 4 * i;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = P[mt][4 * p + e][jj];
        *reinterpret_cast<f32x4*>(rowp + 64 * p) = o;
      }
    }
  if (sact) {
    // the four waves' partial S through LDS (the images are dead): wave 0 adds them in wave order and stores.
    // S tile (mt, nt): register jj of lane (i, kq) is S[basis NT (4 kq + jj) + mt][basis NT i + nt]; mirrored below the diagonal
    __syncthreads();
    f32x4* ex = reinterpret_cast<f32x4*>(smem);
    if (wv > 0) {
#pragma unroll
      for (int t = 0; t < 10; ++t) ex[((wv - 1) * 10 + t) * 64 + lane] = S[t];
    }
    __syncthreads();
    if (wv == 0) {
      int t = 0;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = mt; nt < NT; ++nt) {
          const f32x4 v = ((S[t] + ex[(0 * 10 + t) * 64 + lane]) + ex[(1 * 10 + t) * 64 + lane]) + ex[(2 * 10 + t) * 64 + lane];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int r = NT * (4 * kq + jj) + mt, c = NT * i + nt;
            base[(int64_t)r * ldp + np + c] = v[jj];
            if (nt != mt) base[(int64_t)c * ldp + np + r] = v[jj];
          }
          ++t;
        }
    }
  }
}

// ---- C = A B^T for long contractions: A [rows][lda] (kdim columns, kdim % 256 == 0), B [64][ldb], C [rows][ldc] -----------------
constexpr int ROWGEMM_CHUNK_GB = 8;                   // 16-row blocks of a wave whose accumulators live in registers
constexpr size_t rowgemm_chunk_smem_bytes() { return (size_t)(4 * 64 * 64 + 4 * 4 * 1024) * sizeof(float); }   // B chunk + 4 V images

// nblk (16-row blocks) a multiple of 4 GB; group g = blocks [4 GB g, 4 GB (g + 1)), wave w takes GB consecutive ones.
__global__ __launch_bounds__(256, 1) void k_rowgemm_chunk(const float* __restrict__ A, int64_t lda, int kdim,
                                                          const float* __restrict__ B, int64_t ldb,
                                                          float* __restrict__ C, int64_t ldc, int ngroups) {
  constexpr int NT = 4, KP = 64, GB = ROWGEMM_CHUNK_GB;
  using namespace pmf_chunk;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sH = smem;                                   // [4 panels][KP][64], swizzled rows (row 16 nt + i = basis NT i + nt)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  float* sV = smem + 4 * KP * 64 + wv * 4096;         // this wave's [4 panels][16][64]
  const int nchunk = kdim >> 8;
  unsigned vo[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * q + (lane >> 4);
    vo[q] = (unsigned)(row * (int)lda * 4 + 16 * ((lane & 15) ^ vx(row)));
  }
  const char* Ab = reinterpret_cast<const char*>(A);
  // V piece (block blk, chunk c): request q of panel p
  auto issue_v = [&](int64_t blk, int c, int p, int q) {
    PMF_GLDS16(Ab + (size_t)blk * (16 * lda * 4) + (size_t)c * 1024 + p * 256 + vo[q], sV + p * 1024 + q * 256);
  };
  const int drow = lane >> 4, dchunk = lane & 15;
  auto issue_h = [&](int c) {                         // this wave's quarter of B's chunk c: rows 4 rg .. 4 rg + 3, rg = 4 u + wv
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int rg = 4 * u + wv, row = 4 * rg + drow;
        const int bas = NT * (row & 15) + (row >> 4);
        PMF_GLDS16(B + (size_t)bas * ldb + 256 * c + 64 * p + 4 * (dchunk ^ (row & 15)), sH + p * (KP * 64) + rg * 256);
      }
  };
  auto wg_barrier = [&]() {                           // (leaves LDS-DMA requests in flight)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t blk0 = (int64_t)g * (4 * GB) + wv * GB;
    f32x4 acc[GB][NT];
#pragma unroll
    for (int bb = 0; bb < GB; ++bb)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[bb][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the first piece's panels 0-2 (panel 3 goes out under the first steps, as every block's does)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) issue_v(blk0, 0, p, q);
    for (int c = 0; c < nchunk; ++c) {
      wg_barrier();                                   // every wave has finished with the chunk in sH
      issue_h(c);
      wait_vm<0>();
      wg_barrier();                                   // ... and the new one has landed, all four quarters
#pragma unroll
      for (int bb = 0; bb < GB; ++bb) {
        const int64_t blk = blk0 + bb;
        // the piece behind this one: next block of the chunk, or block 0 of the next chunk; none behind the group's last
        // (the group's very last piece re-requests its own panels behind their last read -- harmless, and the counts of the
        //  waits below stay those of straight-line code)
        const bool last = bb == GB - 1;
        const bool more = !(last && c + 1 == nchunk);
        const int64_t nblk_ = !more ? blk : last ? blk0 : blk + 1;
        const int nc = !more ? c : last ? c + 1 : c;
        f32x4 fa[2], fb[2][NT];
        auto load_step = [&](int s, int buf) {
          const int p = s >> 2, chunk = 4 * (s & 3) + kq;
          fa[buf] = *reinterpret_cast<const f32x4*>(sV + p * 1024 + voff(i, chunk));
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) fb[buf][nt] = lds_read4(sH + p * (KP * 64), 16 * nt + i, chunk);
        };
        // steady order of requests: [this block's panel 3 under steps 0-3][next piece's panels 0, 1, 2 under steps 4-15]
        if (bb == 0) wait_vm<0>(); else wait_vm<8>(); // panel 0 landed (younger: panels 1, 2); behind a chunk switch everything has
        load_step(0, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
          if (s + 1 < 16) {
            if (((s + 1) & 3) == 0) wait_vm<7>();     // next panel landed: 4 requests of a younger panel + the 3 of this panel's steps so far
            load_step(s + 1, (s + 1) & 1);
          }
          const int buf = s & 1;
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[bb][nt] = mfma16(fa[buf][e], fb[buf][nt][e], acc[bb][nt]);
          if (s < 4) issue_v(blk, c, 3, s);
          else issue_v(nblk_, nc, (s >> 2) - 1, s & 3);
          if (s + 1 < 16) {
#pragma unroll
            for (int q = 0; q < NT + 1; ++q) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // ---- the group's rows out: bases NT i .. NT i + 3 of row 4 kq + j as one 16-byte store ----
#pragma unroll
    for (int bb = 0; bb < GB; ++bb) {
      float* dst = C + ((blk0 + bb) * 16 + 4 * kq) * ldc + NT * i;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(dst + (int64_t)j * ldc) = f32x4{acc[bb][0][j], acc[bb][1][j], acc[bb][2][j], acc[bb][3][j]};
    }
  }
}
