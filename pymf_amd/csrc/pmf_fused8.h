// pmf_fused8.h -- the one-pass NMF iteration for 64 < num_bases <= 128 (NT = 8 tiles of bases).
//
// pmf_fused.h keeps H (k x n) and G = H H^T (k x k) in LDS and the whole P = W^T V (k x n)
// accumulator of a wave's rows in registers; at k = 128, n = 256 that is 128 + 64 KiB of LDS and 512
// accumulator registers per wave -- neither exists.  Here the four waves of a workgroup COOPERATE on
// 64-row tiles and split the work differently in the two halves of a tile:
//   phase A (row split)   wave w owns rows 16w .. 16w+15:  Num = V_b H^T (K = n), Den = W_b G (K = 128),
//                         epilogue W_b <- (W_b * Num) / (Den + 1e-9)          pymf/nmf.py:128-132
//                         -> new rows to HBM (16-byte stores) and, transposed, into an LDS tile W^T;
//   phase B (base split)  wave w owns bases 32w .. 32w+31:  P[32 x n] += W_tile^T V_tile over all 64
//                         rows (the four V images in LDS), and 9 of the 36 upper tiles of S += W^T W
//                         (pymf/nmf.py:124-125; the reference updates W before H, nmf.py:183-187)
// so a wave holds P for 32 bases only (128 accumulator registers at n = 256) and no cross-wave sum
// is needed at the end.  H and G do not fit into LDS next to the V tile: their B-operand fragments
// come straight from L2 (global_load_dwordx4 in fragment layout, 16 bytes per lane, one step ahead of
// the MFMAs; the four waves read the same addresses at about the same time, so most of it hits the
// CU's L1).  LDS: V tile 64 KiB (4 images of 16 x n), old-W images 32 KiB, W^T tile 32 KiB = 128 KiB.
//
// Order inside a tile: phase A -> epilogue -> barrier -> P part of phase B -> barrier (V images free)
// -> LDS-DMA of the NEXT tile's V and old-W images is issued -> S part of phase B (needs W^T only;
// ~4.6 k cycles, it hides the DMA's HBM latency: vmcnt counts in order, so a DMA issued right before
// phase A would stall phase A's first fragment wait) -> barrier (W^T free).
#pragma once
#include "pmf_fused.h"

template <int NPANEL>
constexpr size_t fused8_smem_bytes() {
  return (size_t)(4 * NPANEL * 1024 + 4 * 2 * 1024 + 4 * 16 * 128) * sizeof(float);
}

// old-W image: [16 rows][128 floats], 16-byte chunk c (0..31) of row r stored at chunk c ^ vtile_xor(r)
__device__ __forceinline__ int wtile_off(int row, int chunk) { return row * 128 + ((chunk ^ vtile_xor(row)) << 2); }
// W^T tile: [128 bases][64 rows], 16-byte chunk c (0..15 = 4 rows) of basis b stored at chunk c ^ g(b)
__device__ __forceinline__ int wt_key(int b) { return ((b & 15) ^ (b >> 3)) & 15; }
__device__ __forceinline__ int wt_off(int b, int chunk) { return b * 64 + ((chunk ^ wt_key(b)) << 2); }

template <int NPANEL, int MODE>
__global__ __launch_bounds__(256, 1) void k_nmf_fused8(const float* __restrict__ V, float* __restrict__ W,
                                                        const float* __restrict__ H,
                                                        const float* __restrict__ G, int tile_per,
                                                        int tile_extra, float lamb,
                                                        float* __restrict__ slab,
                                                        const int* __restrict__ stop
#ifdef PMF_STAMPS
                                                        , unsigned long long* __restrict__ dbg
#endif
                                                        ) {
  constexpr int KP = 128;
  constexpr int NP = 64 * NPANEL, NTP = 4 * NPANEL;
  constexpr int NSN = 4 * NPANEL;      // Num steps (one 16-byte k-group each)
  constexpr int NSA = NSN + 8;         // + Den steps (K = 128)
  constexpr int LA = 4;                // fragment look-ahead (steps)
  static_assert(MODE != FUSED_SNMF, "SNMF iterates in Gram space");
  if (stop != nullptr && *stop != 0) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sVall = smem;                              // 4 row blocks x [NPANEL][16][64]
  float* sWn = sVall + 4 * NPANEL * 1024;           // new W tile, same image form: 4 row blocks x [2][16][64]
  float* sWoall = sWn + 4 * 2 * 1024;               // 4 row blocks x [16][128]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  float* sV = sVall + wv * (NPANEL * 1024);         // the images this wave FILLS (rows 16 wv ..)
  float* sWo = sWoall + wv * (16 * 128);

  const int b = blockIdx.x;
  const int t0 = b * tile_per + (b < tile_extra ? b : tile_extra);
  const int ntile = tile_per + (b < tile_extra ? 1 : 0);

  // LDS-DMA geometry (per-lane byte offsets relative to a scalar row base, as in pmf_fused.h)
  unsigned voff[4], woff[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * q + (lane >> 4);
    const int c = (lane & 15) ^ vtile_xor(row);
    voff[q] = (unsigned)(row * NP * 4 + 16 * c);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {                     // one instruction = 2 rows x 512 B of the old-W image
    const int row = 2 * q + (lane >> 5);
    const int c = (lane & 31) ^ vtile_xor(row);
    woff[q] = (unsigned)(row * KP * 4 + 16 * c);
  }
  const char* Vb = reinterpret_cast<const char*>(V);
  const char* Wb = reinterpret_cast<const char*>(W);
  // DMA d (0 .. 8 + 4 NPANEL - 1) of this wave's 16 rows of the tile: old W first, then the V panels
  auto issue_dma = [&](int tile, int d) {
    const size_t r0 = (size_t)tile * 64 + 16 * wv;
    if (d < 8) PMF_GLDS16(Wb + r0 * (KP * 4) + woff[d], sWo + d * 256);
    else {
      const int p = (d - 8) >> 2, q = (d - 8) & 3;
      PMF_GLDS16(Vb + r0 * (NP * 4) + p * 256 + voff[q], sV + p * 1024 + q * 256);
    }
  };
  constexpr int NDMA = 8 + 4 * NPANEL;

  // Base split: this wave owns bases 32 wv .. 32 wv + 31 in EVERY product of the tile.  Column i of its
  // base tile bt is basis 32 wv + 2 i + bt (a lane's two tiles are two consecutive bases: 8-byte W accesses).
  f32x4 P[2][NTP];       // P[bt][4 p + e]:  rows = own bases, columns {64 p + 4 c + e}
  f32x4 S[2][8];         // S[bt][4 q + e]:  rows = own bases, columns = bases {64 q + 4 c + e}
#pragma unroll
  for (int bt = 0; bt < 2; ++bt) {
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) P[bt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) S[bt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // B-operand fragments straight from global memory (L2 / L1): H and G (symmetric) are both read along
  // the row of the basis: 16 contiguous bytes per lane, 2 loads per step.
  const float* Hrow = H + (size_t)(32 * wv + 2 * i) * NP + 4 * kq;
  const float* Grow = G + (size_t)(32 * wv + 2 * i) * KP + 4 * kq;
  f32x4 fb[LA][2];
  auto bload = [&](int s) {
    if (s < NSN) {
      fb[s % LA][0] = *reinterpret_cast<const f32x4*>(Hrow + 16 * s);
      fb[s % LA][1] = *reinterpret_cast<const f32x4*>(Hrow + NP + 16 * s);
    } else {
      fb[s % LA][0] = *reinterpret_cast<const f32x4*>(Grow + 16 * (s - NSN));
      fb[s % LA][1] = *reinterpret_cast<const f32x4*>(Grow + KP + 16 * (s - NSN));
    }
  };

  if (ntile > 0) {
#pragma unroll
    for (int d = 0; d < NDMA; ++d) issue_dma(t0, d);
  }
#ifdef PMF_STAMPS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts6 = 0;
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int tt = 0; tt < ntile; ++tt) {
    const int tile = t0 + tt;
    const bool more = tt + 1 < ntile;
    PMF_STAMP(ts0);
    // the first fragments do not depend on the tile: requested before the images are waited for
#pragma unroll
    for (int s = 0; s < LA; ++s) bload(s);
    wait_vmcnt<2 * LA>();             // every DMA of this tile (older than the 2 LA loads) has landed
    __syncthreads();                  // ... for all four waves; the new-W tile of the last tile is free
    PMF_STAMP(ts1);

    // ---------------- phase A (own 32 bases, all 64 rows): Num = V H^T, Den = W G ----------------
    f32x4 num[4][2], den[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int bt = 0; bt < 2; ++bt) {
        num[rb][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
        den[rb][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    f32x4 fa[2][4];
    auto aload = [&](int s, f32x4 (&dst)[4]) {
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
        if (s < NSN) dst[rb] = vtile_read4(sVall + rb * (NPANEL * 1024) + (s >> 2) * 1024, i, 4 * (s & 3) + kq);
        else dst[rb] = *reinterpret_cast<const f32x4*>(sWoall + rb * (16 * 128) + wtile_off(i, 4 * (s - NSN) + kq));
      }
    };
    aload(0, fa[0]);
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
      if (s + 1 < NSA) aload(s + 1, fa[(s + 1) & 1]);
      const int buf = s & 1;
      f32x4 b0 = fb[s % LA][0], b1 = fb[s % LA][1];
      if (s < NSN) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
            num[rb][0] = mfma16(fa[buf][rb][e], b0[e], num[rb][0]);
            num[rb][1] = mfma16(fa[buf][rb][e], b1[e], num[rb][1]);
          }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) {
            den[rb][0] = mfma16(fa[buf][rb][e], b0[e], den[rb][0]);
            den[rb][1] = mfma16(fa[buf][rb][e], b1[e], den[rb][1]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s + LA < NSA) bload(s + LA);        // into the slot step s has just released
      __builtin_amdgcn_sched_barrier(0);
    }
    PMF_STAMP(ts2);

    // ---------------- epilogue + P part, row block by row block ----------------
    // W <- (W * Num) / (Den + eps) for 64 rows x own 32 bases
    // row block by row block (8 elements per lane each); the division in stages as in pmf_fused.h
    f32x4 wn[4][2];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      float wold[2][4], tnum[2][4], dd[2][4], rr[2][4], qq[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* src = sWoall + rb * (16 * 128) + wtile_off(4 * kq + j, 8 * wv + (i >> 1)) + 2 * (i & 1);
        wold[0][j] = src[0];
        wold[1][j] = src[1];
      }
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float w0 = wold[bt][j];
          if (MODE == FUSED_BNMF) {                              // bnmf.py:87-90, W *= W1 / W2
            tnum[bt][j] = num[rb][bt][j] + (3.0f * lamb) * (w0 * w0);
            dd[bt][j] = ((den[rb][bt][j] + (2.0f * lamb) * (w0 * w0 * w0)) + lamb * w0) + PMF_EPS_DEN;
          } else if (MODE == FUSED_RNMF) {                       // rnmf.py:109-115 on D = S - data, no epsilon
            const float x = num[rb][bt][j];
            tnum[bt][j] = fabsf(x) - x;
            dd[bt][j] = 2.0f * den[rb][bt][j];
          } else {
            tnum[bt][j] = w0 * num[rb][bt][j];                   // nmf.py:131 (multiply first)
            dd[bt][j] = den[rb][bt][j] + PMF_EPS_DEN;
          }
        }
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[bt][j] = __builtin_amdgcn_rcpf(dd[bt][j]);
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int j = 0; j < 4; ++j) qq[bt][j] = tnum[bt][j] * rr[bt][j];
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int j = 0; j < 4; ++j) tnum[bt][j] = fmaf(-dd[bt][j], qq[bt][j], tnum[bt][j]);   // residual
#pragma unroll
      for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float w = fmaf(tnum[bt][j], rr[bt][j], qq[bt][j]);        // pmf_div (nmf.py:132)
          if (MODE == FUSED_BNMF) w = wold[bt][j] * w;
          if (MODE == FUSED_RNMF) w = dd[bt][j] != 0.f ? wold[bt][j] * w : 0.f;   // 0/0 on the zero padding
          wn[rb][bt][j] = w;
        }
      // new rows: to HBM (8 bytes per lane, 128 contiguous bytes per row) and into the new-W tile
      float* wdst = W + ((size_t)tile * 64 + 16 * rb + 4 * kq) * KP + 32 * wv + 2 * i;
      float* ldst = sWn + rb * 2048 + (wv >> 1) * 1024;           // panel = basis / 64
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 pr = {wn[rb][0][j], wn[rb][1][j]};
        *reinterpret_cast<f32x2*>(wdst + j * KP) = pr;
        *reinterpret_cast<f32x2*>(ldst + vtile_off(4 * kq + j, 8 * (wv & 1) + (i >> 1)) + 2 * (i & 1)) = pr;
      }
      // ---- phase B, P part of this row block: P += W_new^T V.  The new rows are still in registers
      // (register j of lane group q IS row 4 q + j of the A operand); its 128 MFMAs run beside the VALU
      // of the NEXT row block's epilogue ----
      {
        const float* sVr = sVall + rb * (NPANEL * 1024);
#pragma unroll
        for (int p = 0; p < NPANEL; ++p)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 bf = vtile_read4(sVr + p * 1024, 4 * kq + j, i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              P[0][4 * p + e] = mfma16(wn[rb][0][j], bf[e], P[0][4 * p + e]);
              P[1][4 * p + e] = mfma16(wn[rb][1][j], bf[e], P[1][4 * p + e]);
            }
          }
      }
    }
    PMF_STAMP(ts3);
    PMF_STAMP(ts4);
    __syncthreads();                  // new-W tile complete; every wave is done with the V and old-W images
    PMF_STAMP(ts5);

    // ---------------- phase B, S part: S += W_new^T W_new (own bases x all bases), the DMA of the
    // next tile's images spread over its first half (two per 8 MFMAs) ----------------
    {
      int d = 0;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 bf = vtile_read4(sWn + rb * 2048 + q * 1024, 4 * kq + j, i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              S[0][4 * q + e] = mfma16(wn[rb][0][j], bf[e], S[0][4 * q + e]);
              S[1][4 * q + e] = mfma16(wn[rb][1][j], bf[e], S[1][4 * q + e]);
            }
            // two per step in the first half of the part: the last one is then >= 5 k cycles old when
            // the next tile waits for it
            if (more && d < NDMA) issue_dma(tile + 1, d);
            if (more && d + 1 < NDMA) issue_dma(tile + 1, d + 1);
            d += 2;
          }
    }
    PMF_STAMP(ts6);
#ifdef PMF_STAMPS
    acc[0] += ts1 - ts0; acc[1] += ts2 - ts1; acc[2] += ts3 - ts2; acc[3] += ts4 - ts3; acc[4] += ts5 - ts4;
    acc[5] += ts6 - ts5;
#endif
  }
#ifdef PMF_STAMPS
  if (dbg && lane == 0) {
    unsigned long long* dd_ = dbg + ((size_t)blockIdx.x * 4 + wv) * 9;
    for (int q = 0; q < 8; ++q) dd_[q] = acc[q];
    dd_[8] = (unsigned long long)ntile;
  }
#endif

  // ---- slab: tile-major; P tile (2 wv + bt, nt), then S tile (2 wv + bt, ct): no cross-wave sum ----
  constexpr int NTU = 8 * NTP + 64;
  f32x4* out = reinterpret_cast<f32x4*>(slab) + (size_t)blockIdx.x * NTU * 64 + lane;
#pragma unroll
  for (int bt = 0; bt < 2; ++bt) {
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) out[((2 * wv + bt) * NTP + nt) * 64] = P[bt][nt];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) out[(8 * NTP + (2 * wv + bt) * 8 + ct) * 64] = S[bt][ct];
  }
}

// Slabs of k_nmf_fused8: block t sums tile t of every slab (float64, fixed order) and scatters it into
// the row-major (P | S) buffer.  Tile (g = 2 w + bt, .): tile row m is basis 32 w + 2 m + bt; P tile
// (g, nt = 4 p + e) holds columns 64 p + 4 c + e (lane c), S tile (g, ct = 4 q + e) bases 64 q + 4 c + e.
__global__ __launch_bounds__(1024) void k_reduce_slabs_tiles8(const float* __restrict__ slab, int nslabs,
                                                              int NTP, int np, float* __restrict__ out,
                                                              const int* __restrict__ stop) {
  __shared__ double part[16][64][4];
  if (stop != nullptr && *stop != 0) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  constexpr int KP = 128;
  const int NTU = 8 * NTP + 64;
  const int tile = blockIdx.x;
  const f32x4* p = reinterpret_cast<const f32x4*>(slab) + (size_t)tile * 64 + lane;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
  for (int c = wv; c < nslabs; c += 16) {
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
    const bool isP = tile < 8 * NTP;
    const int g = isP ? tile / NTP : (tile - 8 * NTP) / 8;
    const int ct = isP ? tile % NTP : (tile - 8 * NTP) % 8;
    const int row = 32 * (g >> 1) + 2 * (4 * kq + r) + (g & 1);
    const int col = 64 * (ct >> 2) + 4 * i + (ct & 3);
    out[(int64_t)row * ldp + (isP ? 0 : np) + col] = v;
  }
}

#ifndef PMF_FUSED_KERNEL_ONLY
static inline bool fused8_shape_ok(int NT, int np) { return NT == 8 && np % 64 == 0 && np >= 64 && np <= 256; }

static inline int fused8_grid_for(int64_t mp) {
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  return (int)std::min<int64_t>(mp / 64, cus);
}

template <int NPANEL, int MODE>
static int launch_fused8_t(hipStream_t s, const float* V, float* W, const float* H, const float* G, int64_t mp,
                           int wgs, float lamb, float* slab, const int* stop) {
  const int ntiles = (int)(mp / 64);
  const size_t smem = fused8_smem_bytes<NPANEL>();
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nmf_fused8<NPANEL, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return PMF_EHIP;
    attr_done = true;
  }
  hipLaunchKernelGGL((k_nmf_fused8<NPANEL, MODE>), dim3(wgs), dim3(256), smem, s, V, W, H, G, ntiles / wgs,
                     ntiles % wgs, lamb, slab, stop);
  return PMF_OK;
}

static inline int launch_fused8(hipStream_t s, int mode, int np, const float* V, float* W, const float* H,
                                const float* G, int64_t mp, int wgs, float lamb, float* slab, const int* stop) {
#define PMF_FUSED8_CASE(B)                                                                              \
  case B:                                                                                               \
    return mode == FUSED_BNMF   ? launch_fused8_t<B, FUSED_BNMF>(s, V, W, H, G, mp, wgs, lamb, slab, stop) \
           : mode == FUSED_RNMF ? launch_fused8_t<B, FUSED_RNMF>(s, V, W, H, G, mp, wgs, lamb, slab, stop) \
                                : launch_fused8_t<B, FUSED_NMF>(s, V, W, H, G, mp, wgs, lamb, slab, stop);
  switch (np / 64) {
    PMF_FUSED8_CASE(1)
    PMF_FUSED8_CASE(2)
    PMF_FUSED8_CASE(3)
    PMF_FUSED8_CASE(4)
  }
#undef PMF_FUSED8_CASE
  return PMF_EINVAL;
}
#endif  // PMF_FUSED_KERNEL_ONLY
