// pmf_fused.h -- ONE pass over V per NMF iteration (the bench's dominant kernel).
//
// For every 16-row block b of V (rows r0..r0+15), one wave does, back to back:
//   phase A   Num = V_b H^T  (K = n)          Den = W_b G  (G = H H^T, K = k)
//   epilogue  W_b <- (W_b * Num) / (Den + 1e-9)              pymf/nmf.py:128-132
//   phase B   P += W_b^T V_b (k x n)          S += W_b^T W_b (k x k)   (new W_b;
//             partials of pymf/nmf.py:124-125, legal because the reference updates
//             W before H, nmf.py:183-187)
// so V is read from HBM once per iteration and W once (read) + once (write).
//
// Waves are autonomous (no barrier inside the loop): H (k x n) and G (k x k) sit
// read-only in LDS for the whole workgroup; each wave owns a private LDS image of
// its current V block (16 x n) and W block (16 x k), filled by LDS-DMA
// (global_load_lds_dwordx4, source-side XOR swizzle) and drained with counted
// s_waitcnt vmcnt(N).  The epilogue's accumulator registers (C layout) ARE the A
// operand of phase B: MFMA step j of a 16-row block contracts rows {4q + j}, which
// is exactly what register j of lane group q holds -- no LDS round trip for W_b.
// Phase B walks the column panels in the same order phase A does, so panel p of the
// NEXT block is DMA-issued the moment phase B has finished with panel p: a single
// LDS buffer gives a full block of prefetch distance.
//
// MODE = FUSED_SNMF builds the semi-NMF W step into the same pass (pymf/snmf.py:67-70), reassociated:
// W_b = (V_b H^T) inv(H H^T) = V_b M^T with M^T = inv(H H^T) H formed beforehand in float64
// (k_snmf_mt): the kernel is handed M^T in the place of H, phase A's product IS the new W block
// (already in the accumulator layout phase B wants), there is no second small product, no old W is
// read and G is not used.
//
// Accumulators: P = NT x 4*NPANEL tiles and S = NT x NT tiles of 16x16 (4 VGPRs
// each) stay in registers for the wave's whole row range; one wave per SIMD
// (__launch_bounds__(256, 1)) so the 512-entry unified VGPR/AGPR file holds them.
#pragma once
#include "pmf_dev.h"
#include <hip/hip_ext.h>
#include "../../include/pymf_hip.h"
#include "pmf_fused_api.h"   // FUSED_*, FusedCtl

#define PMF_GLDS16(gsrc, ldst)                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc), \
                                   (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)

// Diagnostic build only (-DPMF_STAMPS): per-section cycle sums of one wave (guide section 7,
// "In-kernel stamps").  Never compiled into the shipped library.
#ifdef PMF_STAMPS
#define PMF_STAMP(var)                                                           \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)
#else
#define PMF_STAMP(var) do { } while (0)
#endif


// Slab / W-row stores.  A/B switches (diagnostic builds): PMF_SLAB_SC1 / PMF_SLAB_NT, PMF_W_NT.
__device__ __forceinline__ void slab_store16(f32x4* p, f32x4 v) {
#if defined(PMF_SLAB_SC1)
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif defined(PMF_SLAB_NT)
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ void wrow_store(float* p, float v) {
#if defined(PMF_W_NT)
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// SPLIT = 2 (wide data): the two waves of a pair share every 16-row block, each owning NPANEL of
// the 2 * NPANEL column panels; H holds all panels, and a 4 * NT KiB exchange area carries the
// partial Num tiles between the partners.
template <int NT, int NPANEL, int SPLIT = 1>
constexpr size_t fused_smem_bytes() {
  return (size_t)64 * (SPLIT * NPANEL * 16 * NT + 16 * NT + 4 * 16 * NPANEL + 4 * 16) * sizeof(float) +
         (SPLIT == 2 ? (size_t)4 * NT * 1024 : 0);
}

// Swizzle of the 16-row V tile.  It is read two ways: phase A takes 16 rows x one chunk per
// k-group (swz_off's pattern), phase B takes ONE row per k-group x all 16 chunks.  A
// ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... -- two
// k-groups per LDS pass -- so rows r and r+4 must not put chunks {0-3,12-15} and {4-11} on the
// same slots: flipping bit 3 of the XOR key on rows with bit 2 set keeps both reads
// conflict-free (phase A only needs the key to be a bijection that preserves r ^ r' == 1).
__device__ __forceinline__ int vtile_xor(int row) { return (row ^ ((row & 4) << 1)) & 15; }
__device__ __forceinline__ int vtile_off(int row, int chunk) { return row * 64 + ((chunk ^ vtile_xor(row)) << 2); }
__device__ __forceinline__ f32x4 vtile_read4(const float* base, int row, int chunk) {
  return *reinterpret_cast<const f32x4*>(base + vtile_off(row, chunk));
}

// blk_per / blk_extra: 16-row blocks per wave (floor) and the number of waves that take one
// more; computed on the host so every loop bound and base address is scalar (SGPR).
template <int NT, int NPANEL, int MODE, int SPLIT = 1>
__global__ __launch_bounds__(256, 1) void k_nmf_fused(const float* __restrict__ V,
                                                       float* __restrict__ W,
                                                       const float* __restrict__ H,
                                                       const float* __restrict__ G, int blk_per,
                                                       int blk_extra, float lamb,
                                                       float* __restrict__ slab,
                                                       const FusedCtl ctl, int ngp
#ifdef PMF_STAMPS
                                                       , unsigned long long* __restrict__ dbg
#endif
                                                       ) {
  constexpr int KP = 16 * NT;
  static_assert(SPLIT == 1 || (SPLIT == 2 && MODE != FUSED_SNMF), "SPLIT: 1, or 2 for the NMF/BNMF epilogues");
  constexpr int NPT = NPANEL * SPLIT;   // column panels of the data; a wave owns NPANEL of them
  constexpr int NP = 64 * NPT;
  constexpr int NTP = 4 * NPANEL;   // column tiles of P held by one wave
  // DMA of the next block's V panel p is spread over phase B's panel p+1 steps and the last
  // panel over the next phase A's first steps (one LDS-DMA per 16 MFMAs) -- needs 4 panels.
  constexpr bool SPREAD = (NPANEL == 4);
  constexpr bool SNMF = (MODE == FUSED_SNMF);
  // free-running loops (pmf_factorize): a launch enqueued behind a converged iteration is a no-op
  if (ctl.stop != nullptr && *ctl.stop != 0) return;
#ifdef PMF_STAMPS
  unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
  const unsigned long long rt0 = wall_clock64();     // 100 MHz constant-rate counter: comparable across CUs and with the host's events
  PMF_STAMP(tk0);
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sH = smem;                             // [NPT][KP][64]   swizzled rows
  float* sG = sH + NPT * KP * 64;               // [KP][64]
  float* sVall = sG + KP * 64;                  // 4 waves x [NPANEL][16][64]
  float* sWall = sVall + 4 * NPANEL * 16 * 64;  // 4 waves x [16][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (SGPR)
  const int i = lane & 15, kq = lane >> 4;
  float* sV = sVall + wv * (NPANEL * 1024);
  float* sW = sWall + wv * 1024;
  f32x4* sX = reinterpret_cast<f32x4*>(sWall + 4 * 1024);   // SPLIT 2: [4 waves][NT][64] partial Num
  const int half = SPLIT == 2 ? (wv & 1) : 0;   // which NPANEL panels of a block this wave owns
  const int hp = half * NPANEL;

  // ---- this wave's (SPLIT 2: this pair's) contiguous range of 16-row blocks (all scalar) ----
  const int gw = SPLIT == 2 ? blockIdx.x * 2 + (wv >> 1) : blockIdx.x * 4 + wv;
  const int b0 = gw * blk_per + (gw < blk_extra ? gw : blk_extra);
  const int nb = blk_per + (gw < blk_extra ? 1 : 0);
  const int nb_wg = blk_per + (blk_extra > 0 ? 1 : 0);   // SPLIT 2: barrier trips, same for every wave
  // Up to 16 bases the kernel is HBM-bound (4.5-4.9 TB/s), and there it pays to INTERLEAVE the blocks: wave gw takes blocks
  // gw, gw + nw, gw + 2 nw, ... -- at any moment the chip works on nw adjacent blocks, a window that sweeps through V and
  // W -- instead of a contiguous range of its own (1 048 576 x 384, k = 16: 0.350 -> 0.333 ms; x 256: 0.237 -> 0.234; the
  // MFMA-bound instantiations do not care: 1 048 576 x 256, k = 64 0.6199 / 0.6184, the 131 072-row shard 0.1035 / 0.1039,
  // cfg2 0.0593 / 0.0594 -- tools/interleave_ab.py, profiles/r04_experiments.md -- and keep their summation order).
  constexpr bool interleave = NT == 1;
  const int nw_all = (int)gridDim.x * (SPLIT == 2 ? 2 : 4);
  const int bfirst = interleave ? gw : b0;
  const int bstep = interleave ? nw_all : 1;

  // LDS-DMA geometry: one instruction = 4 rows x 256 B; lane L fills physical chunk (L & 15)
  // of row 4q + (L >> 4), so it fetches logical chunk (L & 15) ^ row.  Per-lane BYTE offsets
  // (32-bit) relative to a scalar row base: the source address is SGPR base + VGPR offset.
  unsigned voff[4], woff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * q + (lane >> 4);
    const int c = (lane & 15) ^ vtile_xor(row);   // the W image is read both ways as well (Den fragments, old W rows)
    voff[q] = (unsigned)(row * NP * 4 + 16 * c);
    woff[q] = (unsigned)(row * KP * 4 + 16 * (4 * c < KP ? c : 0));   // beyond k: valid, never read
  }
  const char* Vb = reinterpret_cast<const char*>(V);
  const char* Wb = reinterpret_cast<const char*>(W);
  auto issue_v = [&](int blk, int p, int q) {      // V rows of block blk, panel p, DMA q
#ifdef PMF_ABLATE_DMA      // timing-only diagnostic build: outputs are wrong
    if (blk != bfirst) return;
#endif
    PMF_GLDS16(Vb + ((size_t)blk * (16 * NP * 4) + (hp + p) * 256) + voff[q], sV + p * 1024 + q * 256);
  };
  auto issue_w = [&](int blk, int q) {
#ifdef PMF_ABLATE_DMA
    if (blk != bfirst) return;
#endif
    PMF_GLDS16(Wb + (size_t)blk * (16 * KP * 4) + woff[q], sW + q * 256);
  };

  f32x4 P[NT][NTP];
  f32x4 S[NT][NT];       // only nt >= mt is accumulated (S is symmetric)
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) P[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) S[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- prologue: H, G and the first block's V / W tiles into LDS by LDS-DMA, in the order the first block uses them ----
  // Round 4: the requests go out stage by stage -- [H panel p of all waves' shares][V panel p of the first block] (W behind
  // stage 0), G last -- and the first block waits for ITS stage only (counted vmcnt + a workgroup barrier per stage, H being
  // shared): its MFMAs start when a quarter of the 72-90 KiB a workgroup pulls has landed instead of behind all of it
  // (rounds 1-3: one vmcnt(0) + barrier; 6.5 k of the 97 k cycles of the kernel at 65 536 x 512, k = 32).
  // The MFMA N index of a lane is free: column i of tile nt of Num / Den (and of the new W) is
  // basis NT i + nt, i.e. the NT tiles of a lane hold NT CONSECUTIVE bases, so the old W rows are
  // fetched and the new ones stored NT floats at a time (16-byte accesses at k = 64).  All it takes
  // is to put the rows of H and G into the LDS images in that order; k_reduce_slabs_tiles knows it.
  constexpr int HP = SPLIT * NT;                 // H pieces of a stage per wave (KP / 4 row groups per panel over 4 waves)
  constexpr int GP = SNMF ? 0 : NT;              // G pieces per wave
  constexpr int WP = SNMF ? 0 : 4;               // W pieces per wave
  {
    const int drow = lane >> 4, dchunk = lane & 15;
#pragma unroll
    for (int st = 0; st < NPANEL; ++st) {
#pragma unroll
      for (int hh = 0; hh < SPLIT; ++hh)
#pragma unroll
        for (int u = 0; u < NT; ++u) {                           // H: panel p, rows 4rg..4rg+3
          const int p = hh * NPANEL + st, rg = 4 * u + wv;
          const int row = 4 * rg + drow;                         // LDS row 16 nt + i ...
          const int bas = NT * (row & 15) + (row >> 4);          // ... holds basis NT i + nt
          const float* src = H + (size_t)bas * NP + 64 * p + 4 * (dchunk ^ (row & 15));
          PMF_GLDS16(src, sH + p * (KP * 64) + rg * 256);
        }
      if (nb > 0 && (st < NPANEL - 1 || !SPREAD)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_v(bfirst, st, q);
      }
      if (st == 0 && !SNMF && nb > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_w(bfirst, q);
      }
    }
#pragma unroll
    for (int u = 0; u < (SNMF ? 0 : NT); ++u) {                  // G: rows 4rg..4rg+3 (SNMF: no G)
      const int rg = 4 * u + wv;
      const int row = 4 * rg + drow;
      const int bas = NT * (row & 15) + (row >> 4);
      int c = dchunk ^ (row & 15);
      if (4 * c >= KP) c = 0;                                  // beyond k: valid, never read
      if (ngp == 0) {
        PMF_GLDS16(G + bas * KP + 4 * c, sG + rg * 256);
      } else {
        // G arrives as the ngp per-workgroup partial sums of k_nmf_h_gram: add them here (fixed
        // order) instead of making that kernel wait for its last workgroup
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < ngp; ++w)
          g += *reinterpret_cast<const f32x4*>(G + (size_t)w * KP * KP + bas * KP + 4 * c);
        *reinterpret_cast<f32x4*>(sG + rg * 256 + drow * 64 + dchunk * 4) = g;
      }
    }
  }
  if (ctl.stop != nullptr && ctl.conv_iter >= 0) {   // the previous iteration's error and convergence test
    double t0 = ctl.tt[0], t1 = ctl.tt[1];
    for (int q = 1; q < ctl.ntt; ++q) { t0 += ctl.tt[2 * q]; t1 += ctl.tt[2 * q + 1]; }
    const double e2 = ctl.vnorm2 - 2.0 * t0 + t1;
    int st = 0;
    if (!(e2 > 1e-3 * ctl.vnorm2)) {
      st = 2;
    } else {
      const double f = sqrt(e2);
      if (blockIdx.x == 0 && tid == 0) ctl.ferr[ctl.conv_iter] = f;
      if (ctl.conv_iter > 1 && fabs(f - ctl.ferr[ctl.conv_iter - 1]) / ctl.nsamp < ctl.eps) st = 1;
    }
    if (st != 0) {                                   // uniform over the grid: W has not been touched
      if (blockIdx.x == 0 && tid == 0) { ctl.stop[1] = ctl.conv_iter; ctl.stop[0] = st; }
      return;
    }
  }
  // workgroup barrier that leaves LDS-DMA pieces in flight (__syncthreads() may drain them: vmcnt(0))
  auto wg_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // Stage st of the prologue has landed in every wave's share.  Younger than this wave's V pieces of stage st: the later
  // stages' H and V pieces, G, W behind stage 0 and (SPREAD) the last panel's pieces issued under steps 0-2 of the block.
  auto first_stage = [&](int st) {
    if (st >= NPANEL - 1) wait_vmcnt<0>();            // the last stage: G and W with it (the Den steps follow)
    else if (SPREAD) {
      if (st == 0) wait_vmcnt<WP + 2 * (HP + 4) + HP + GP>();
      else if (st == 1) wait_vmcnt<(HP + 4) + HP + GP + 3>();
      else wait_vmcnt<HP + GP + 4>();                 // (the four last-panel pieces have gone out under steps 0-3)
    } else {
      if (st == 0) wait_vmcnt<WP + (NPANEL - 1) * (HP + 4) + GP>();
      else if (st == 1) wait_vmcnt<(NPANEL > 2 ? (NPANEL - 2) * (HP + 4) : 0) + GP>();
      else if (st == 2) wait_vmcnt<(NPANEL > 3 ? (NPANEL - 3) * (HP + 4) : 0) + GP>();
      else if (st == 3) wait_vmcnt<(NPANEL > 4 ? (NPANEL - 4) * (HP + 4) : 0) + GP>();
      else wait_vmcnt<0>();
    }
    wg_barrier();
  };
  if (nb == 0) {          // a wave without a block: its shares of H and G, and the first block's barriers
    wait_vmcnt<0>();
#pragma unroll
    for (int st = 0; st < NPANEL; ++st) wg_barrier();
  }

  // Fragment double buffers: step s+1's LDS reads are issued before step s's MFMAs.
  f32x4 fa[2];
  f32x4 fb[2][NT];
  f32x4 bf[2];
  f32x4 wp[NT];          // previous block's new W rows: its S MFMAs run under this block's epilogue
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wp[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int NSN = 4 * NPANEL;      // Num steps (one 16-byte k-group each)
  constexpr int NSA = SNMF ? NSN : NSN + NT;   // + Den steps (SNMF: phase A's product is the new W itself)

#ifdef PMF_STAMPS
  PMF_STAMP(tk1);
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0;
  unsigned long long acc_wait = 0, acc_a = 0, acc_dma = 0, acc_epi = 0, acc_b = 0;
#endif
  for (int b = 0; b < (SPLIT == 2 ? nb_wg : nb); ++b) {
    if (SPLIT == 2) {
      wg_barrier();                                     // the exchange area is free again
      if (b >= nb) { wg_barrier(); continue; }          // a pair without this block only keeps step (nb == 0: above)
    }
    const int blk = bfirst + b * bstep;
    const bool more = (b + 1 < nb);
    PMF_STAMP(ts0);

    f32x4 num[NT], den[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      num[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      den[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // step s < NSN: Num k-group (panel s/4, t = s%4); s >= NSN: Den k-group t = s - NSN
    auto load_step = [&](int s, int buf) {
      if (s < NSN) {
        const int p = s >> 2, chunk = 4 * (s & 3) + kq;
        fa[buf] = vtile_read4(sV + p * 1024, i, chunk);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[buf][nt] = lds_read4(sH + (hp + p) * (KP * 64), 16 * nt + i, chunk);
      } else {
        const int chunk = 4 * (s - NSN) + kq;
        fa[buf] = vtile_read4(sW, i, chunk);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[buf][nt] = lds_read4(sG, 16 * nt + i, chunk);
      }
    };
    // V panel p of this block has landed.  vmcnt counts every later VMEM op in issue order.
    //  clumped:  panels p+1.. (4 DMA each) may still be in flight.
    //  SPREAD :  order per block is [last panel x4 under phase A steps 0..3][W' x4][stores]
    //            [V' p0 x4][V' p1 x4][V' p2 x4]; see the wait at each use below.
    auto wait_panel = [&](int p) {
      if (SPREAD) {
        if (p == 0) wait_vmcnt<8>();        // V p1, p2 of this block still allowed in flight
        else if (p == 1) wait_vmcnt<7>();   // p2 (4) + last-panel DMAs issued after steps 0,1,2
        else if (p == 2) wait_vmcnt<4>();   // the 4 last-panel DMAs
        else wait_vmcnt<0>();
      } else {
        if (p == 0) wait_vmcnt<4 * (NPANEL - 1)>();
        else if (p == 1) wait_vmcnt<(NPANEL > 1 ? 4 * (NPANEL - 2) : 0)>();
        else if (p == 2) wait_vmcnt<(NPANEL > 2 ? 4 * (NPANEL - 3) : 0)>();
        else wait_vmcnt<0>();
      }
    };

    // ---------------- phase A: Num = V_b H^T, Den = W_b G ----------------
    const bool first = (b == 0);                    // the first block runs behind the prologue's stages
    if (first) first_stage(0);
    else wait_panel(0);          // also covers the (older) W image
    PMF_STAMP(ts1);
    load_step(0, 0);
#pragma unroll
    for (int s = 0; s < NSA; ++s) {
      if (s + 1 < NSA) {
        if (s + 1 < NSN && ((s + 1) & 3) == 0) {
          if (first) first_stage((s + 1) >> 2);
          else wait_panel((s + 1) >> 2);
        }
        load_step(s + 1, (s + 1) & 1);
      }
      const int buf = s & 1;
      if (s < NSN) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) num[nt] = mfma16(fa[buf][e], fb[buf][nt][e], num[nt]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) den[nt] = mfma16(fa[buf][e], fb[buf][nt][e], den[nt]);
      }
      if (SPREAD && s < 4) issue_v(blk, NPANEL - 1, s);   // this block's last panel, 1 DMA/step
      // issue order inside the step: one LDS read (of step s+1) per 2 MFMAs (of step s), so a
      // read's issue slot hides under an executing MFMA and the last read is >= 6 MFMAs old when
      // the next step needs it; the DMA goes last
      if (s + 1 < NSA) {
#pragma unroll
        for (int g = 0; g < NT + 1; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, NT >= 4 ? 2 : 1, 0);   // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  // DS read
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);                     // remaining MFMAs
      __builtin_amdgcn_sched_barrier(0);
    }
    if (SPLIT == 2) {
      // Num of this wave's panels + Num of the partner's = V_b H^T: both partners form the same sum
      // (a + b == b + a exactly), hence the same new W rows
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) sX[(wv * NT + nt) * 64 + lane] = num[nt];
      wg_barrier();
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) num[nt] += sX[((wv ^ 1) * NT + nt) * 64 + lane];
    }
    PMF_STAMP(ts2);
    // old W rows in the accumulator (C) layout, then the W image is free: prefetch the next block's
    float wold[NT][4];
    if (!SNMF) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = 4 * kq + j, col = NT * i + nt;       // NT consecutive floats per (lane, j)
          wold[nt][j] = sW[vtile_off(row, col >> 2) + (col & 3)];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (more) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_w(blk + bstep, q);
      }
    }
    PMF_STAMP(ts3);

    // ------- epilogue W_b <- (W_b * Num) / (Den + eps), interleaved with S += of the previous block -------
    f32x4 wn[NT];
    float* wdst = W + (size_t)blk * (16 * KP) + (4 * kq) * KP + NT * i;
    if (MODE != FUSED_SNMF) {
      // The division of the 4 NT elements in STAGES (all numerators, all reciprocals, all quotients,
      // all residuals, all corrections): a wave issues in order, so an element-by-element chain of
      // dependent VALU ops also holds up the S MFMAs queued behind it; stage by stage every
      // instruction has 4 NT - 1 independent ones between itself and its consumer.
      float tnum[NT][4], dd[NT][4], rr[NT][4], qq[NT][4];
      constexpr int NSM = 4 * (NT * (NT + 1) / 2);           // S MFMAs of the previous block
      auto s_mfmas = [&](int lo, int hi) {                   // ... numbers lo .. hi-1 of them
        if (SPLIT == 1 || half == 0) {
          int q = 0;                                         // j outermost: consecutive MFMAs hit different tiles
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int mt = 0; mt <= nt; ++mt) {
                if (q >= lo && q < hi) S[mt][nt] = mfma16(wp[mt][j], wp[nt][j], S[mt][nt]);
                ++q;
              }
        }
      };
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float w0 = wold[nt][j];
          if (MODE == FUSED_BNMF) {                              // bnmf.py:87-90, W *= W1 / W2
            tnum[nt][j] = num[nt][j] + (3.0f * lamb) * (w0 * w0);
            dd[nt][j] = ((den[nt][j] + (2.0f * lamb) * (w0 * w0 * w0)) + lamb * w0) + PMF_EPS_DEN;
          } else if (MODE == FUSED_RNMF) {                       // rnmf.py:109-115 on D = S - data, no epsilon
            const float x = num[nt][j];
            tnum[nt][j] = fabsf(x) - x;
            dd[nt][j] = 2.0f * den[nt][j];
          } else {
            tnum[nt][j] = w0 * num[nt][j];                       // nmf.py:131 (multiply first)
            dd[nt][j] = den[nt][j] + PMF_EPS_DEN;
          }
        }
      s_mfmas(0, NSM / 5);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[nt][j] = __builtin_amdgcn_rcpf(dd[nt][j]);
      s_mfmas(NSM / 5, 2 * NSM / 5);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) qq[nt][j] = tnum[nt][j] * rr[nt][j];
      s_mfmas(2 * NSM / 5, 3 * NSM / 5);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) tnum[nt][j] = fmaf(-dd[nt][j], qq[nt][j], tnum[nt][j]);   // residual
      s_mfmas(3 * NSM / 5, 4 * NSM / 5);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float w = fmaf(tnum[nt][j], rr[nt][j], qq[nt][j]);        // pmf_div (nmf.py:132)
          if (MODE == FUSED_BNMF) w = wold[nt][j] * w;
          if (MODE == FUSED_RNMF) w = dd[nt][j] != 0.f ? wold[nt][j] * w : 0.f;   // 0/0 on the zero padding
          wn[nt][j] = w;
#ifndef PMF_ABLATE_WSTORE
          if (SPLIT == 1 || half == 0) wrow_store(&wdst[j * KP + nt], w);
#endif
        }
      s_mfmas(4 * NSM / 5, NSM);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float w = num[nt][j];                        // W = V (inv(H H^T) H)^T, snmf.py:67-70 reassociated
          wn[nt][j] = w;
          wrow_store(&wdst[j * KP + nt], w);
#pragma unroll
          for (int mt = 0; mt <= nt; ++mt) S[mt][nt] = mfma16(wp[mt][j], wp[nt][j], S[mt][nt]);
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wp[nt] = wn[nt];

    PMF_STAMP(ts4);
    // ---------------- phase B: P += W_b^T V_b ----------------
    // The MFMA column index of a lane is free: P tile (mt, 4p + e) holds columns {64p + 4i + e},
    // so ONE 16-byte read of chunk i of row 4kq + j feeds the four column tiles of a panel step
    // (k_reduce_slabs_tiles undoes the permutation when it scatters into the row-major buffer).
    auto load_bf = [&](int s, int buf) {
      const int p = s >> 2, row = 4 * kq + (s & 3);
      bf[buf] = vtile_read4(sV + p * 1024, row, i);
    };
    load_bf(0, 0);
#pragma unroll
    for (int s = 0; s < NSN; ++s) {
      if (s + 1 < NSN) load_bf(s + 1, (s + 1) & 1);
      const int p = s >> 2, j = s & 3, buf = s & 1;
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          P[mt][4 * p + nt] = mfma16(wn[mt][j], bf[buf][nt], P[mt][4 * p + nt]);
      if (SPREAD) {
        // panel p-1 was fully read one panel ago (its last reads fed step 4p-1's MFMAs)
        if (p >= 1 && more) issue_v(blk + bstep, p - 1, j);
      }
      if (s + 1 < NSN) {
        __builtin_amdgcn_sched_group_barrier(0x008, NT >= 2 ? 2 : 1, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                    // the step's one DS read
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!SPREAD && j == 3) {
        // every read of panel p has returned: refill it for the next block
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) {
#pragma unroll
          for (int q = 0; q < 4; ++q) issue_v(blk + bstep, p, q);
        }
      }
    }
    if (SPREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // last panel's reads returned
#ifdef PMF_STAMPS
    PMF_STAMP(ts5);
    acc_wait += ts1 - ts0; acc_a += ts2 - ts1; acc_dma += ts3 - ts2; acc_epi += ts4 - ts3; acc_b += ts5 - ts4;
#endif
  }
#ifdef PMF_STAMPS
  PMF_STAMP(tk2);
#endif
  // S of the last block
  if (SPLIT == 1 || half == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt <= nt; ++mt) S[mt][nt] = mfma16(wp[mt][j], wp[nt][j], S[mt][nt]);
  }

  // ---- sum the 4 waves' accumulators through LDS, write ONE tile-major slab per workgroup ----
  // Slab = NTU tiles of 64 lanes x float4 in the accumulator layout (P tiles, then the S tiles
  // on/above the diagonal): every store is a coalesced 1-KiB b128 wave store, spread over all 4
  // waves.  k_reduce_slabs_tiles sums the slabs and scatters to the row-major (P | S) buffer,
  // mirroring S.
  constexpr int NTU = NT * NTP + NT * (NT + 1) / 2;
  __syncthreads();
  f32x4* ex = reinterpret_cast<f32x4*>(smem);   // two regions of NTU*64 f32x4
  static_assert((size_t)2 * NTU * 64 * 16 <= fused_smem_bytes<NT, NPANEL, SPLIT>(), "exchange fits");
  auto put = [&](int region) {
    f32x4* dst = ex + (size_t)region * NTU * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NTP; ++nt) dst[(mt * NTP + nt) * 64] = P[mt][nt];
#pragma unroll
      for (int nt = mt; nt < NT; ++nt)
        dst[(NT * NTP + mt * NT - (mt * (mt - 1)) / 2 + (nt - mt)) * 64] = S[mt][nt];
    }
  };
  auto add = [&](int region) {
    const f32x4* src = ex + (size_t)region * NTU * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NTP; ++nt) P[mt][nt] += src[(mt * NTP + nt) * 64];
#pragma unroll
      for (int nt = mt; nt < NT; ++nt)
        S[mt][nt] += src[(NT * NTP + mt * NT - (mt * (mt - 1)) / 2 + (nt - mt)) * 64];
    }
  };
  if (wv >= 2) put(wv - 2);
  __syncthreads();
  if (wv < 2) add(wv);
  __syncthreads();
  if (wv < 2) put(wv);
  __syncthreads();
  // (round 5, tools/r05_ab.sh: a tail in which every tile is finished by ONE wave -- the others hand it over through LDS once, three
  //  rounds over two buffers, slab stores beside the next round's LDS traffic, same summation order and bits -- measured 84.5 -> 83.9 us
  //  on the 131 072-row shard and 594.0 -> 596.0 us at 1 048 576 rows: the tail is bound by the 18.5 MiB of slab stores all 256
  //  workgroups issue at the same moment, not by its LDS traffic or its three barriers.  Not kept.)
  if (SPLIT == 1) {
    f32x4* out = reinterpret_cast<f32x4*>(slab) + (size_t)blockIdx.x * NTU * 64 + lane;
    for (int t = wv; t < NTU; t += 4)
      slab_store16(&out[t * 64], ex[t * 64 + lane] + ex[(size_t)NTU * 64 + t * 64 + lane]);
  } else {
    // region h holds the P tiles of panels h * NPANEL .. (summed over the two pairs); S lives in
    // region 0.  The slab is tile-major over ALL panels: P tile (mt, 4 * panel + e), then S.
    constexpr int NTPT = 4 * NPT, NTUT = NT * NTPT + NT * (NT + 1) / 2;
    f32x4* out = reinterpret_cast<f32x4*>(slab) + (size_t)blockIdx.x * NTUT * 64 + lane;
    for (int t = wv; t < 2 * NT * NTP; t += 4) {
      const int h = t / (NT * NTP), q = t % (NT * NTP), mt = q / NTP, nt = q % NTP;
      slab_store16(&out[(mt * NTPT + h * NTP + nt) * 64], ex[(size_t)h * NTU * 64 + q * 64 + lane]);
    }
    for (int t = wv; t < NT * (NT + 1) / 2; t += 4)
      slab_store16(&out[(NT * NTPT + t) * 64], ex[(NT * NTP + t) * 64 + lane]);
  }
#ifdef PMF_STAMPS
  PMF_STAMP(tk3);
  if (dbg && lane == 0) {
    unsigned long long* d = dbg + ((size_t)blockIdx.x * 4 + wv) * 12;
    d[0] = acc_wait; d[1] = acc_a; d[2] = acc_dma; d[3] = acc_epi; d[4] = acc_b; d[5] = (unsigned long long)nb;
    d[6] = tk1 - tk0; d[7] = tk3 - tk2;
    d[8] = rt0; d[9] = wall_clock64(); d[10] = tk3 - tk0; d[11] = tk2 - tk1;
  }
#endif
}

// ---- host-side dispatch -----------------------------------------------------------------
#ifndef PMF_FUSED_KERNEL_ONLY
static inline bool fused_shape_ok(int NT, int np) {
  // LDS: 64 * (npanel * 16 NT + 16 NT + 64 npanel + 64) floats <= 160 KiB
  const int npanel = np / 64;
  if (np % 64 != 0 || npanel < 1) return false;
  const int max_panels = NT == 1 ? 6 : NT == 2 ? 5 : NT == 4 ? 4 : 0;
  return npanel <= max_panels;
}

// Wider data, k <= 32: two waves share each 16-row block (SPLIT 2), 2 x 3 or 2 x 4 panels.
static inline bool fused_shape_split(int NT, int np) {
  if (NT == 2) return np == 384 || np == 512;
  if (NT == 1) return np == 512;
  return false;
}

// Workgroups to launch (one per CU), 0 when the shape is not covered by the fused kernel.
static inline int fused_grid_for(int NT, int np, int64_t mp, bool allow_split = true) {
  const bool split = allow_split && !fused_shape_ok(NT, np) && fused_shape_split(NT, np);
  if (!fused_shape_ok(NT, np) && !split) return 0;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    cus = prop.multiProcessorCount;
  const int64_t nblk = mp / 16;
  int64_t wgs = split ? (nblk + 1) / 2 : (nblk + 3) / 4;
  if (wgs > cus) wgs = cus;
  return (int)wgs;
}

static inline const char* fused_kernel_name(int NT, int np, int mode = FUSED_NMF) {
  static char buf[64];
  // (the template arguments as rocprofv3 prints them: NT, column panels PER WAVE, and SPLIT 2 where two waves share a block)
  const bool split = !fused_shape_ok(NT, np) && fused_shape_split(NT, np) && mode != FUSED_SNMF;
  snprintf(buf, sizeof(buf), "k_nmf_fused<%d,%d%s%s>", NT, split ? np / 128 : np / 64,
           mode == FUSED_SNMF ? ",snmf" : mode == FUSED_BNMF ? ",bnmf" : mode == FUSED_RNMF ? ",rnmf" : "", split ? ",SPLIT 2" : "");
  return buf;
}

template <int NT, int NPANEL, int MODE, int SPLIT = 1>
static int launch_fused_t(hipStream_t s, const float* V, float* W, const float* H, const float* G,
                          int64_t mp, int wgs, float lamb, float* slab, const FusedCtl& ctl, int ngp,
                          hipEvent_t e0, hipEvent_t e1) {
  const int nblk = (int)(mp / 16), nw = wgs * (SPLIT == 2 ? 2 : 4);   // waves, or pairs of waves
  const int blk_per = nblk / nw, blk_extra = nblk % nw;
  const size_t smem = fused_smem_bytes<NT, NPANEL, SPLIT>();
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nmf_fused<NT, NPANEL, MODE, SPLIT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return PMF_EHIP;
    attr_done = true;
  }
  // e0 / e1 (pmf_profile_enable): the events ride on the DISPATCH packet itself (hipExtLaunchKernelGGL: start / stop stamps of
  // this kernel, what rocprofv3 reports) instead of two hipEventRecord barrier packets around it -- those cost the loop 6 us per
  // iteration (65 536 x 512, k = 32: 59.3 -> 65.3 us; profiles/r05_experiments.md) and read 5 us long
  if (e0 != nullptr && e1 != nullptr)
    hipExtLaunchKernelGGL((k_nmf_fused<NT, NPANEL, MODE, SPLIT>), dim3(wgs), dim3(256), (uint32_t)smem, s, e0, e1, 0u, V, W, H, G, blk_per,
                          blk_extra, lamb, slab, ctl, ngp);
  else
    hipLaunchKernelGGL((k_nmf_fused<NT, NPANEL, MODE, SPLIT>), dim3(wgs), dim3(256), smem, s, V, W, H, G, blk_per,
                       blk_extra, lamb, slab, ctl, ngp);
  return PMF_OK;
}

// G: H H^T [KP][KP] float32 (NMF, BNMF, RNMF).  FUSED_SNMF: H is M^T = inv(H H^T) H and G is unused.
static inline int launch_fused(hipStream_t s, int mode, int NT, int np, const float* V, float* W,
                               const float* H, const float* G, int64_t mp, int wgs, float lamb,
                               float* slab, const FusedCtl& ctl, int ngp = 0, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr) {
  if (!fused_shape_ok(NT, np) && fused_shape_split(NT, np) && mode != FUSED_SNMF) {
    const int skey = NT * 10 + np / 128;
#define PMF_FUSED_SPLIT_CASE(K, A, B)                                                                   \
  case K:                                                                                               \
    return mode == FUSED_BNMF   ? launch_fused_t<A, B, FUSED_BNMF, 2>(s, V, W, H, G, mp, wgs, lamb, slab, ctl, ngp, e0, e1) \
           : mode == FUSED_RNMF ? launch_fused_t<A, B, FUSED_RNMF, 2>(s, V, W, H, G, mp, wgs, lamb, slab, ctl, ngp, e0, e1) \
                                : launch_fused_t<A, B, FUSED_NMF, 2>(s, V, W, H, G, mp, wgs, lamb, slab, ctl, ngp, e0, e1);
    switch (skey) {
      PMF_FUSED_SPLIT_CASE(14, 1, 4)
      PMF_FUSED_SPLIT_CASE(23, 2, 3)
      PMF_FUSED_SPLIT_CASE(24, 2, 4)
    }
#undef PMF_FUSED_SPLIT_CASE
    return PMF_EINVAL;
  }
  const int key = NT * 10 + np / 64;
#define PMF_FUSED_CASE(K, A, B)                                                                  \
  case K:                                                                                        \
    return mode == FUSED_SNMF   ? launch_fused_t<A, B, FUSED_SNMF>(s, V, W, H, G, mp, wgs, lamb, slab, ctl, ngp, e0, e1) \
           : mode == FUSED_BNMF ? launch_fused_t<A, B, FUSED_BNMF>(s, V, W, H, G, mp, wgs, lamb, slab, ctl, ngp, e0, e1) \
           : mode == FUSED_RNMF ? launch_fused_t<A, B, FUSED_RNMF>(s, V, W, H, G, mp, wgs, lamb, slab, ctl, ngp, e0, e1) \
                                : launch_fused_t<A, B, FUSED_NMF>(s, V, W, H, G, mp, wgs, lamb, slab, ctl, ngp, e0, e1);
  switch (key) {
    PMF_FUSED_CASE(11, 1, 1)
    PMF_FUSED_CASE(12, 1, 2)
    PMF_FUSED_CASE(13, 1, 3)
    PMF_FUSED_CASE(14, 1, 4)
    PMF_FUSED_CASE(15, 1, 5)
    PMF_FUSED_CASE(16, 1, 6)
    PMF_FUSED_CASE(21, 2, 1)
    PMF_FUSED_CASE(22, 2, 2)
    PMF_FUSED_CASE(23, 2, 3)
    PMF_FUSED_CASE(24, 2, 4)
    PMF_FUSED_CASE(25, 2, 5)
    PMF_FUSED_CASE(41, 4, 1)
    PMF_FUSED_CASE(42, 4, 2)
    PMF_FUSED_CASE(43, 4, 3)
    PMF_FUSED_CASE(44, 4, 4)
  }
#undef PMF_FUSED_CASE
  return PMF_EINVAL;
}
#endif  // PMF_FUSED_KERNEL_ONLY
