// pmf_coop.h -- the one-pass NMF iteration for the shapes pmf_fused.h cannot hold in LDS / registers:
//     64 < num_bases <= 128 with n <= 384,   and   32 < num_bases <= 64 with 256 < n <= 512.
//
// pmf_fused.h keeps H (k x n) and G = H H^T (k x k) in LDS and the whole P = W^T V (k x n)
// accumulator of a wave's rows in registers; at k = 128, n = 256 that is 128 + 64 KiB of LDS and 512
// accumulator registers per wave, at k = 64, n = 1024 it is 256 KiB and 1 024 -- neither exists.  Here
// the four waves of a workgroup COOPERATE on tiles of 16 RB rows (RB = 4 or 2 row blocks) and EVERY
// product of the tile is split by bases: wave w owns the 16 BT bases 16 BT w .. (BT = 2 for k <= 128,
// BT = 1 for k <= 64) in
//   phase A   Num = V H^T (K = n) and Den = W G (K = KP) for all rows of the tile,
//   epilogue  W <- (W * Num) / (Den + 1e-9)                                   pymf/nmf.py:128-132
//             -> new rows to HBM and into an LDS tile of the new W,
//   phase B   P[16 BT x n] += W_new^T V (the new rows are still in registers: register j of lane group
//             q IS row 4 q + j of the MFMA A operand) and S[16 BT x KP] += W_new^T W_new, whose B
//             operand is the new-W tile in the V images' own form: "KP / 64 more column panels"
//             (pymf/nmf.py:124-125; the reference updates W before H, nmf.py:183-187)
// so a wave holds P for 16 BT bases only and no cross-wave sum is needed at the end.  H and G do not
// fit into LDS next to the V tile: their B-operand fragments come straight from L2
// (global_load_dwordx4 in fragment layout, 16 bytes per lane along the basis row -- G is symmetric, so
// it is read along rows too), BT loads per 4 RB BT MFMAs, requested LA = 4 steps ahead; a scheduling
// fence keeps the compiler from sinking the loads to their uses (the waits are vmcnt(N > 0)).
// LDS: V tile RB x n x 64 B, old-W images and new-W tile RB x 16 x KP x 4 B each.
//
// Order inside a tile: [wait for this tile's images] barrier -> phase A -> epilogue + P part, row block
// by row block -> barrier (new-W tile complete, V and old-W images free) -> S part, with the LDS-DMA of
// the NEXT tile's images issued beside its first MFMAs (vmcnt counts in order: a DMA issued right
// before phase A would stall phase A's first fragment wait).
#pragma once
#include "pmf_fused.h"

template <int BT, int RB, int NPANEL>
constexpr size_t coop_smem_bytes() {
  return (size_t)(RB * NPANEL * 1024 + RB * BT * 1024 + RB * 16 * 64 * BT) * sizeof(float);
}

// old-W image: [16 rows][KP floats], 16-byte chunk c of row r stored at chunk c ^ vtile_xor(r)
template <int KP>
__device__ __forceinline__ int wo_off(int row, int chunk) { return row * KP + ((chunk ^ vtile_xor(row)) << 2); }

template <int BT, int RB, int NPANEL, int MODE>
__global__ __launch_bounds__(256, 1) void k_nmf_coop(const float* __restrict__ V, float* __restrict__ W,
                                                      const float* __restrict__ H,
                                                      const float* __restrict__ G, int tile_per,
                                                      int tile_extra, float lamb,
                                                      float* __restrict__ slab,
                                                      const int* __restrict__ stop
#ifdef PMF_STAMPS
                                                      , unsigned long long* __restrict__ dbg
#endif
                                                      ) {
  constexpr int KP = 64 * BT;          // bases, padded: 4 waves x 16 BT
  constexpr int KT = KP / 16;          // column tiles of S, Den steps
  constexpr int NP = 64 * NPANEL, NTP = 4 * NPANEL;
  constexpr int NSN = 4 * NPANEL;      // Num steps (one 16-byte k-group each)
  constexpr int NSA = NSN + KT;        // + Den steps (K = KP)
  constexpr int LA = 4;                // fragment look-ahead (steps)
  constexpr int NWO = KP / 16;         // DMA instructions per old-W image (1 KiB each)
  constexpr int CPB = NWO + 4 * NPANEL;   // DMA instructions per row block
  constexpr int NDMA = RB * CPB / 4;      // ... per wave and tile
  static_assert(MODE != FUSED_SNMF, "SNMF iterates in Gram space");
  static_assert(RB == 4 || (RB == 2 && CPB % 2 == 0), "RB: 4, or 2 with the row block's DMAs split over two waves");
  if (stop != nullptr && *stop != 0) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sVall = smem;                              // RB row blocks x [NPANEL][16][64]
  float* sWn = sVall + RB * NPANEL * 1024;          // new W tile, same image form: RB row blocks x [BT][16][64]
  float* sWoall = sWn + RB * BT * 1024;             // RB row blocks x [16][KP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;

  const int b = blockIdx.x;
  const int t0 = b * tile_per + (b < tile_extra ? b : tile_extra);
  const int ntile = tile_per + (b < tile_extra ? 1 : 0);

  // LDS-DMA geometry (per-lane byte offsets relative to a scalar row base, as in pmf_fused.h)
  unsigned voff[4], woff[NWO];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * q + (lane >> 4);
    const int c = (lane & 15) ^ vtile_xor(row);
    voff[q] = (unsigned)(row * NP * 4 + 16 * c);
  }
#pragma unroll
  for (int q = 0; q < NWO; ++q) {                   // one instruction = 1 KiB = 256 / KP rows of the old-W image
    constexpr int LPR = KP / 4;                     // lanes per row
    const int row = (64 / LPR) * q + lane / LPR;
    const int c = (lane % LPR) ^ vtile_xor(row);
    woff[q] = (unsigned)(row * KP * 4 + 16 * c);
  }
  const char* Vb = reinterpret_cast<const char*>(V);
  const char* Wb = reinterpret_cast<const char*>(W);
  // DMA r (0 .. CPB - 1) of row block rb: the old-W image first, then the V panels.  r is a compile-time
  // constant at every call site (unrolled loops), so the offset registers are picked statically.
  auto issue_r = [&](int tile, int rb, int r) {
    const size_t r0 = (size_t)tile * (16 * RB) + 16 * rb;
    if (r < NWO) {
      PMF_GLDS16(Wb + r0 * (KP * 4) + woff[r < NWO ? r : 0], sWoall + rb * (16 * KP) + r * 256);
    } else {
      const int p = (r - NWO) >> 2, q = (r - NWO) & 3;
      PMF_GLDS16(Vb + r0 * (NP * 4) + p * 256 + voff[q], sVall + rb * (NPANEL * 1024) + p * 1024 + q * 256);
    }
  };
  // DMA d (0 .. NDMA - 1) of this wave.  RB = 4: wave w fills row block w.  RB = 2: waves w and w + 2 share
  // row block w & 1, the first taking the first half of its DMAs.
  auto issue_dma = [&](int tile, int d) {
    if (RB == 4) {
      issue_r(tile, wv, d);
    } else {
      if ((wv >> 1) == 0) issue_r(tile, wv & 1, d);
      else issue_r(tile, wv & 1, CPB / 2 + d);
    }
  };
  // Base split: this wave owns bases 16 BT wv .. in EVERY product of the tile.  Column i of its base tile
  // bt is basis 16 BT wv + BT i + bt (a lane's BT tiles are BT consecutive bases).
  f32x4 P[BT][NTP];      // P[bt][4 p + e]:  rows = own bases, columns {64 p + 4 c + e}
  f32x4 S[BT][KT];       // S[bt][4 q + e]:  rows = own bases, columns = bases {64 q + 4 c + e}
#pragma unroll
  for (int bt = 0; bt < BT; ++bt) {
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) P[bt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < KT; ++nt) S[bt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int base0 = 16 * BT * wv + BT * i;          // first basis of this lane
  const float* Hrow = H + (size_t)base0 * NP + 4 * kq;
  const float* Grow = G + (size_t)base0 * KP + 4 * kq;
  f32x4 fb[LA][BT];
  auto bload = [&](int s) {
    if (s < NSN) {
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) fb[s % LA][bt] = *reinterpret_cast<const f32x4*>(Hrow + bt * NP + 16 * s);
    } else {
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) fb[s % LA][bt] = *reinterpret_cast<const f32x4*>(Grow + bt * KP + 16 * (s - NSN));
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
    wait_vmcnt<BT * LA>();            // every DMA of this tile (older than the BT LA loads) has landed
    __syncthreads();                  // ... for all four waves; the new-W tile of the last tile is free
    PMF_STAMP(ts1);

    // ---------------- phase A (own bases, all rows of the tile): Num = V H^T, Den = W G ----------------
    f32x4 num[RB][BT], den[RB][BT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int bt = 0; bt < BT; ++bt) {
        num[rb][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
        den[rb][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    f32x4 fa[2][RB];
    auto aload = [&](int s, f32x4 (&dst)[RB]) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if (s < NSN) dst[rb] = vtile_read4(sVall + rb * (NPANEL * 1024) + (s >> 2) * 1024, i, 4 * (s & 3) + kq);
        else dst[rb] = *reinterpret_cast<const f32x4*>(sWoall + rb * (16 * KP) + wo_off<KP>(i, 4 * (s - NSN) + kq));
      }
    };
    aload(0, fa[0]);
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
      if (s + 1 < NSA) aload(s + 1, fa[(s + 1) & 1]);
      const int buf = s & 1;
      if (s < NSN) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) num[rb][bt] = mfma16(fa[buf][rb][e], fb[s % LA][bt][e], num[rb][bt]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) den[rb][bt] = mfma16(fa[buf][rb][e], fb[s % LA][bt][e], den[rb][bt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s + LA < NSA) bload(s + LA);        // into the slot step s has just released
      __builtin_amdgcn_sched_barrier(0);
    }
    PMF_STAMP(ts2);

    // ---------------- epilogue + P part, row block by row block ----------------
    // W <- (W * Num) / (Den + eps) for the tile's rows x own bases; the division in stages as in
    // pmf_fused.h (rcp, quotient, residual, correction)
    f32x4 wn[RB][BT];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      float wold[BT][4], tnum[BT][4], dd[BT][4], rr[BT][4], qq[BT][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* src = sWoall + rb * (16 * KP) + wo_off<KP>(4 * kq + j, base0 >> 2) + (base0 & 3);
#pragma unroll
        for (int bt = 0; bt < BT; ++bt) wold[bt][j] = src[bt];
      }
#pragma unroll
      for (int bt = 0; bt < BT; ++bt)
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
      for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[bt][j] = __builtin_amdgcn_rcpf(dd[bt][j]);
#pragma unroll
      for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int j = 0; j < 4; ++j) qq[bt][j] = tnum[bt][j] * rr[bt][j];
#pragma unroll
      for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int j = 0; j < 4; ++j) tnum[bt][j] = fmaf(-dd[bt][j], qq[bt][j], tnum[bt][j]);   // residual
#pragma unroll
      for (int bt = 0; bt < BT; ++bt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float w = fmaf(tnum[bt][j], rr[bt][j], qq[bt][j]);        // pmf_div (nmf.py:132)
          if (MODE == FUSED_BNMF) w = wold[bt][j] * w;
          if (MODE == FUSED_RNMF) w = dd[bt][j] != 0.f ? wold[bt][j] * w : 0.f;   // 0/0 on the zero padding
          wn[rb][bt][j] = w;
        }
      // new rows: to HBM (BT floats per lane) and into the new-W tile (panel = basis / 64)
      float* wdst = W + ((size_t)tile * (16 * RB) + 16 * rb + 4 * kq) * KP + base0;
      float* ldst = sWn + rb * (BT * 1024) + (base0 >> 6) * 1024;
      const int lchunk = (base0 & 63) >> 2, loff = base0 & 3;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* l2 = ldst + vtile_off(4 * kq + j, lchunk) + loff;
        if (BT == 2) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const f32x2 pr = {wn[rb][0][j], wn[rb][BT - 1][j]};
          *reinterpret_cast<f32x2*>(wdst + j * KP) = pr;
          *reinterpret_cast<f32x2*>(l2) = pr;
        } else {
          wdst[j * KP] = wn[rb][0][j];
          l2[0] = wn[rb][0][j];
        }
      }
      // ---- phase B, P part of this row block: P += W_new^T V.  The new rows are still in registers
      // (register j of lane group q IS row 4 q + j of the A operand) ----
      {
        const float* sVr = sVall + rb * (NPANEL * 1024);
#pragma unroll
        for (int p = 0; p < NPANEL; ++p)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 bf = vtile_read4(sVr + p * 1024, 4 * kq + j, i);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int bt = 0; bt < BT; ++bt) P[bt][4 * p + e] = mfma16(wn[rb][bt][j], bf[e], P[bt][4 * p + e]);
          }
      }
    }
    PMF_STAMP(ts3);
    PMF_STAMP(ts4);
    __syncthreads();                  // new-W tile complete; every wave is done with the V and old-W images
    PMF_STAMP(ts5);

    // ---------------- phase B, S part: S += W_new^T W_new (own bases x all bases); the DMA of the
    // next tile's images beside its first MFMAs (two per 4 BT MFMAs) ----------------
    {
      int d = 0;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int q = 0; q < BT; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 bf = vtile_read4(sWn + rb * (BT * 1024) + q * 1024, 4 * kq + j, i);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int bt = 0; bt < BT; ++bt) S[bt][4 * q + e] = mfma16(wn[rb][bt][j], bf[e], S[bt][4 * q + e]);
            if (more && d < NDMA) issue_dma(tile + 1, d);
            if (more && d + 1 < NDMA) issue_dma(tile + 1, d + 1);
            d += 2;
          }
      if (more) {
#pragma unroll
        for (int dd_ = (RB * BT * 4) * 2; dd_ < NDMA; ++dd_) issue_dma(tile + 1, dd_);   // a short S part: the rest at once
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

  // ---- slab: tile-major; P tile (BT wv + bt, nt), then S tile (BT wv + bt, ct): no cross-wave sum ----
  constexpr int NTU = 4 * BT * NTP + 4 * BT * KT;
  f32x4* out = reinterpret_cast<f32x4*>(slab) + (size_t)blockIdx.x * NTU * 64 + lane;
#pragma unroll
  for (int bt = 0; bt < BT; ++bt) {
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) out[((BT * wv + bt) * NTP + nt) * 64] = P[bt][nt];
#pragma unroll
    for (int ct = 0; ct < KT; ++ct) out[(4 * BT * NTP + (BT * wv + bt) * KT + ct) * 64] = S[bt][ct];
  }
}

// Slabs of k_nmf_coop: block t sums tile t of every slab (float64, fixed order) and scatters it into
// the row-major (P | S) buffer.  Tile (g = BT w + bt, .): tile row m is basis 16 BT w + BT m + bt; P tile
// (g, nt = 4 p + e) holds columns 64 p + 4 c + e (lane c), S tile (g, ct = 4 q + e) bases 64 q + 4 c + e.
__global__ __launch_bounds__(1024) void k_reduce_slabs_coop(const float* __restrict__ slab, int nslabs,
                                                            int BT, int NTP, int np, float* __restrict__ out,
                                                            const int* __restrict__ stop) {
  __shared__ double part[16][64][4];
  if (stop != nullptr && *stop != 0) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int KP = 64 * BT, KT = KP / 16;
  const int NTU = 4 * BT * NTP + 4 * BT * KT;
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
    const bool isP = tile < 4 * BT * NTP;
    const int g = isP ? tile / NTP : (tile - 4 * BT * NTP) / KT;
    const int ct = isP ? tile % NTP : (tile - 4 * BT * NTP) % KT;
    const int row = 16 * BT * (g / BT) + BT * (4 * kq + r) + (g % BT);
    const int col = 64 * (ct >> 2) + 4 * i + (ct & 3);
    out[(int64_t)row * ldp + (isP ? 0 : np) + col] = v;
  }
}

#ifndef PMF_FUSED_KERNEL_ONLY
// Shapes: (BT, RB, NPANEL) for a context of NT base tiles and np padded columns; false: not covered.
// np must already be padded to the panel count returned in *npanel (coop_pad_np).
static inline bool coop_shape(int NT, int np, int* bt, int* rb, int* npanel) {
  if (np % 64 != 0 || np < 64) return false;
  const int pn = np / 64;
  if (NT == 8) {                       // 64 < k <= 128: n <= 256 on 64-row tiles, n <= 384 on 32-row tiles
    if (pn <= 4) { *bt = 2; *rb = 4; *npanel = pn; return true; }
    if (pn == 6) { *bt = 2; *rb = 2; *npanel = pn; return true; }
    return false;
  }
  if (NT == 4) {                       // 32 < k <= 64, wider than pmf_fused.h takes: 256 < n <= 512
    if (pn == 6 || pn == 8) { *bt = 1; *rb = 4; *npanel = pn; return true; }
    return false;
  }
  return false;
}

// Columns a context should pad to so that the cooperative kernel takes the shape (0: leave as it is).
static inline int coop_pad_np(int NT, int np) {
  const int pn = np / 64;
  if (NT == 8 && pn > 4 && pn <= 6) return 384;
  if (NT == 4 && pn > 4 && pn <= 8) return pn <= 6 ? 384 : 512;
  return 0;
}

static inline int coop_grid_for(int64_t mp, int rb) {
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  return (int)std::min<int64_t>(mp / (16 * rb), cus);
}

template <int BT, int RB, int NPANEL, int MODE>
static int launch_coop_t(hipStream_t s, const float* V, float* W, const float* H, const float* G, int64_t mp,
                         int wgs, float lamb, float* slab, const int* stop) {
  const int ntiles = (int)(mp / (16 * RB));
  const size_t smem = coop_smem_bytes<BT, RB, NPANEL>();
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nmf_coop<BT, RB, NPANEL, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return PMF_EHIP;
    attr_done = true;
  }
  hipLaunchKernelGGL((k_nmf_coop<BT, RB, NPANEL, MODE>), dim3(wgs), dim3(256), smem, s, V, W, H, G, ntiles / wgs,
                     ntiles % wgs, lamb, slab, stop);
  return PMF_OK;
}

static inline int launch_coop(hipStream_t s, int mode, int NT, int np, const float* V, float* W, const float* H,
                              const float* G, int64_t mp, int wgs, float lamb, float* slab, const int* stop) {
  int bt = 0, rb = 0, pn = 0;
  if (!coop_shape(NT, np, &bt, &rb, &pn)) return PMF_EINVAL;
#define PMF_COOP_CASE3(BT_, RB_, PN_)                                                                         \
  if (bt == BT_ && rb == RB_ && pn == PN_)                                                                    \
    return mode == FUSED_BNMF   ? launch_coop_t<BT_, RB_, PN_, FUSED_BNMF>(s, V, W, H, G, mp, wgs, lamb, slab, stop) \
           : mode == FUSED_RNMF ? launch_coop_t<BT_, RB_, PN_, FUSED_RNMF>(s, V, W, H, G, mp, wgs, lamb, slab, stop) \
                                : launch_coop_t<BT_, RB_, PN_, FUSED_NMF>(s, V, W, H, G, mp, wgs, lamb, slab, stop);
#define PMF_COOP_CASE2(BT_, RB_, PN_)                                                                         \
  if (bt == BT_ && rb == RB_ && pn == PN_)                                                                    \
    return mode == FUSED_BNMF ? launch_coop_t<BT_, RB_, PN_, FUSED_BNMF>(s, V, W, H, G, mp, wgs, lamb, slab, stop)   \
                              : launch_coop_t<BT_, RB_, PN_, FUSED_NMF>(s, V, W, H, G, mp, wgs, lamb, slab, stop);
  PMF_COOP_CASE3(2, 4, 1)
  PMF_COOP_CASE3(2, 4, 2)
  PMF_COOP_CASE3(2, 4, 3)
  PMF_COOP_CASE3(2, 4, 4)
  PMF_COOP_CASE2(2, 2, 6)
  PMF_COOP_CASE2(1, 4, 6)
  PMF_COOP_CASE2(1, 4, 8)
#undef PMF_COOP_CASE3
#undef PMF_COOP_CASE2
  return PMF_EINVAL;
}
#endif  // PMF_FUSED_KERNEL_ONLY
