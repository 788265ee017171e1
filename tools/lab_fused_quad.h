// lab_fused_quad.h -- EXPERIMENT (not part of the library; profiles/r04_experiments.md): the one-pass NMF kernel at TWO waves per SIMD, for wide data at k <= 32.
//
// k_nmf_fused<2,4,*,SPLIT 2> (65 536 x 512, k = 32: BASELINE cfg2) holds one wave per SIMD -- its LDS image is exactly
// 160 KiB -- and in that wave's instruction stream everything that is not an MFMA is a hole in the MFMA pipe: the exchange
// of the partial Num tiles between the two waves of a pair (420 cycles per 16-row block), the old W rows (250), the
// division chain of the epilogue (470), the issue of the 16 + 4 LDS-DMA pieces of the next block (about 480):
// 10.7 k cycles per block for 9.1 k cycles of MFMAs (in-kernel stamps, tools/stamp_fused.hip).  A second wave on the SIMD
// fills these holes, and this kernel makes room for it:
//   * a workgroup is 8 waves = 2 QUADS; the four waves of a quad share every 16-row block of the quad's row range, each
//     owning NPANEL of its 4 NPANEL column panels (SPLIT 4), so a wave's V tile (its LDS-DMA target) is half of SPLIT 2's
//     and the eight tiles together take what the four took; ONE H / G image serves all eight waves; the W image (16 x k)
//     is one per quad, each wave fetching a quarter of it;
//   * the quads run HALF A PERIOD APART: a block is an A slot (Num = V_b H^T over the wave's panels, partial Num into the
//     exchange area) and a B slot (sum of the four partial Nums -- every wave of the quad forms the same sum in the same
//     order, hence the same new W rows --, Den = W_b G, the W rule, P += W_b^T V_b over the wave's panels); a workgroup
//     barrier separates the slots, quad 0 is in its A slot while quad 1 is in its B slot and vice versa.  On every SIMD
//     one wave of each quad is resident, so the SIMD always has a dense MFMA stream next to the latency chains of the
//     other quad, and the two barriers per block are met after equal work (A: 64 MFMAs, B: 16 + 3 + 64);
//   * Den is formed in the B slot, behind the barrier: the W image a quarter of which each wave has fetched is complete
//     there, and its 4 NT MFMAs cover the exchange reads;
//   * the W rows are stored and S = W^T W is accumulated a quarter per wave (rows 4q + h of every lane group: MFMA step h),
//     P needs no sum inside a quad (the waves own different columns); at the end quad 1 hands its P tiles to quad 0 through
//     LDS and quad 0 writes the slab from registers -- one LDS pass and one barrier where the four-wave form has two and three.
// Slab layout, H / G images, swizzles, the FusedCtl prologue and the epilogue arithmetic are those of k_nmf_fused (same
// k_reduce_slabs_tiles, same k_nmf_h_gram behind it); reference: pymf/nmf.py:122-132, 183-187.
#pragma once
#include "/root/repo/pymf_amd/csrc/pmf_fused.h"

// Experiment: idle cycles behind every MFMA of the dense streams (see the B slot's comment)
#ifdef PMF_QUAD_PAD
#define PMF_QPAD(acc) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc))
#else
#define PMF_QPAD(acc) do { } while (0)
#endif

template <int NT, int NPANEL>
constexpr size_t quad_smem_bytes() {
  // H [4 NPANEL][KP][64] + G [KP][64] + V 8 waves x [NPANEL][16][64] floats, + X 8 waves x NT KiB
  return (size_t)64 * (4 * NPANEL * 16 * NT + 16 * NT + 8 * NPANEL * 16) * sizeof(float) + (size_t)8 * NT * 1024;
}

template <int NT, int NPANEL, int MODE>
__global__ __launch_bounds__(512, 1) void k_nmf_quad(const float* __restrict__ V, float* __restrict__ W,
                                                      const float* __restrict__ H, const float* __restrict__ G,
                                                      int blk_per, int blk_extra, float lamb, float* __restrict__ slab,
                                                      const FusedCtl ctl, int ngp
#ifdef PMF_STAMPS
                                                      , unsigned long long* __restrict__ dbg
#endif
                                                      ) {
  constexpr int KP = 16 * NT;
  static_assert(MODE != FUSED_SNMF, "NMF / BNMF / RNMF epilogues");
  static_assert(NT <= 2, "k <= 32");
  constexpr int NPT = 4 * NPANEL;   // column panels of the data; a wave owns NPANEL of them
  constexpr int NP = 64 * NPT;
  constexpr int NTP = 4 * NPANEL;   // column tiles of P held by one wave
  constexpr int NS = NT * (NT + 1) / 2;
  if (ctl.stop != nullptr && *ctl.stop != 0) return;
#ifdef PMF_STAMPS
  unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0;
  PMF_STAMP(tk0);
#endif
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sH = smem;                              // [NPT][KP][64]   swizzled rows
  float* sG = sH + NPT * KP * 64;                // [KP][64]
  float* sVall = sG + KP * 64;                   // 8 waves x [NPANEL][16][64]
  f32x4* sX = reinterpret_cast<f32x4*>(sVall + 8 * NPANEL * 16 * 64);   // [8 waves][NT][64] partial Num
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int qd = wv >> 2, h = wv & 3;            // waves w and w + 4 share a SIMD: one wave of each quad
  const int hp = h * NPANEL;
  float* sV = sVall + wv * (NPANEL * 1024);

  const int gw = blockIdx.x * 2 + qd;            // this quad's contiguous range of 16-row blocks (all scalar)
  const int b0 = gw * blk_per + (gw < blk_extra ? gw : blk_extra);
#ifdef PMF_QUAD_SOLO   // diagnostic: quad 1 idles (wrong results)
  const int nb = qd == 0 ? blk_per + (gw < blk_extra ? 1 : 0) : 0;
#else
  const int nb = blk_per + (gw < blk_extra ? 1 : 0);
#endif
  const int nb_wg = blk_per + (blk_extra > 0 ? 1 : 0);

  unsigned voff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * q + (lane >> 4);
    const int c = (lane & 15) ^ vtile_xor(row);
    voff[q] = (unsigned)(row * NP * 4 + 16 * c);
  }
  const char* Vb = reinterpret_cast<const char*>(V);
  auto issue_v = [&](int blk, int p, int q) {
    PMF_GLDS16(Vb + ((size_t)blk * (16 * NP * 4) + (hp + p) * 256) + voff[q], sV + p * 1024 + q * 256);
  };
  // workgroup barrier that leaves the LDS-DMA pieces of the next block in flight (__syncthreads() would drain them: vmcnt(0))
  auto quad_barrier = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  f32x4 P[NT][NTP];
  f32x4 S[NT][NT];       // only nt >= mt is accumulated; this wave's quarter (MFMA step h) of it
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) P[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) S[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (nb > 0) {
#pragma unroll
    for (int p = 0; p < NPANEL; ++p)
#pragma unroll
      for (int q = 0; q < 4; ++q) issue_v(b0, p, q);
  }
  {  // H and G into LDS (whole workgroup, once); row order as in k_nmf_fused: LDS row 16 nt + i holds basis NT i + nt
    const int drow = lane >> 4, dchunk = lane & 15;
    for (int d = wv; d < NPT * (KP / 4); d += 8) {
      const int p = d / (KP / 4), rg = d % (KP / 4);
      const int row = 4 * rg + drow;
      const int bas = NT * (row & 15) + (row >> 4);
      PMF_GLDS16(H + (size_t)bas * NP + 64 * p + 4 * (dchunk ^ (row & 15)), sH + p * (KP * 64) + rg * 256);
    }
    for (int rg = wv; rg < KP / 4; rg += 8) {
      const int row = 4 * rg + drow;
      const int bas = NT * (row & 15) + (row >> 4);
      int c = dchunk ^ (row & 15);
      if (4 * c >= KP) c = 0;
      if (ngp == 0) {
        PMF_GLDS16(G + bas * KP + 4 * c, sG + rg * 256);
      } else {
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < ngp; ++w) g += *reinterpret_cast<const f32x4*>(G + (size_t)w * KP * KP + bas * KP + 4 * c);
        *reinterpret_cast<f32x4*>(sG + rg * 256 + drow * 64 + dchunk * 4) = g;
      }
    }
  }
  if (ctl.stop != nullptr && ctl.conv_iter >= 0) {   // the previous iteration's error and convergence test (as k_nmf_fused)
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
    if (st != 0) {
      if (blockIdx.x == 0 && tid == 0) { ctl.stop[1] = ctl.conv_iter; ctl.stop[0] = st; }
      return;
    }
  }
  wait_vmcnt<0>();
  __syncthreads();

  f32x4 fa[2];
  f32x4 fb[2][NT];
  f32x4 bf[2];
  f32x4 num[NT], den[NT];   // den, wold: formed / fetched in the A slot, used in the B slot
  float wold[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    num[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    den[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) wold[nt][j] = 0.f;
  }
  constexpr int NSN = 4 * NPANEL;      // Num steps (one 16-byte k-group each), and phase B steps
  constexpr int NSA = NSN + NT;        // + Den steps
  constexpr int NWL = NT + 4;          // the A slot's ordinary loads of the old W rows (younger than the block's V pieces)
  // V panel p of the block has landed: younger than its 4 pieces are the later panels' and the NWL loads of W
  auto wait_panel = [&](int p) {
    if (p == 0) wait_vmcnt<4 * (NPANEL - 1) + NWL>();
    else if (p == 1) wait_vmcnt<(NPANEL > 1 ? 4 * (NPANEL - 2) : 0) + NWL>();
    else wait_vmcnt<NWL>();
  };
#ifdef PMF_STAMPS
  PMF_STAMP(tk1);
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0;
  unsigned long long acc_a = 0, acc_x = 0, acc_epi = 0, acc_b = 0, acc_bar = 0, acc_f1 = 0, acc_f2 = 0;
#endif

  for (int t = 0; t <= 2 * nb_wg; ++t) {
    const int tt = t - qd;             // the quad's own slot count: quad 1 runs half a period behind quad 0
    const int b = tt >> 1;
    PMF_STAMP(ts0);
    if (tt >= 0 && b < nb) {
      const int blk = b0 + b;
      if ((tt & 1) == 0) {
        // ================= A slot: Num = V_b H^T over this wave's panels, Den = W_b G =================
        // The old W rows come straight from global memory into registers, in the two layouts they are used in (A operand
        // of Den: row i, one 16-byte k-group; accumulator layout for the rule: rows 4kq + j, bases NT i ..): every wave of
        // the quad fetches the whole 16 x k block (L2 serves three of the four) -- an LDS image shared by the quad would
        // have to be complete before a barrier in front of Den, and Den belongs HERE, at the end of a dense MFMA stream:
        // as a chain of dependent MFMAs at the head of the B slot, beside the other quad's A slot, it crawled (2.9 k cycles
        // for 19 MFMAs, in-kernel stamps).
        const float* wsrc = W + (size_t)blk * (16 * KP);
        f32x4 da[NT];
#pragma unroll
        for (int s = 0; s < NT; ++s) da[s] = *reinterpret_cast<const f32x4*>(wsrc + i * KP + 16 * s + 4 * kq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (NT == 2) {
            const float2 w2 = *reinterpret_cast<const float2*>(wsrc + (4 * kq + j) * KP + NT * i);
            wold[0][j] = w2.x;
            wold[NT - 1][j] = w2.y;
          } else {
            wold[0][j] = wsrc[(4 * kq + j) * KP + NT * i];
          }
        }
        auto load_step = [&](int s, int buf) {
          if (s < NSN) {
            const int p = s >> 2, chunk = 4 * (s & 3) + kq;
            fa[buf] = vtile_read4(sV + p * 1024, i, chunk);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[buf][nt] = lds_read4(sH + (hp + p) * (KP * 64), 16 * nt + i, chunk);
          } else {
            const int chunk = 4 * (s - NSN) + kq;
            fa[buf] = da[s - NSN];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) fb[buf][nt] = lds_read4(sG, 16 * nt + i, chunk);
          }
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          num[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
          den[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        wait_panel(0);
        load_step(0, 0);
#pragma unroll
        for (int s = 0; s < NSA; ++s) {
          if (s + 1 < NSA) {
            if (s + 1 < NSN && ((s + 1) & 3) == 0) wait_panel((s + 1) >> 2);
            if (s + 1 == NSN) wait_vmcnt<0>();            // the old W rows
            load_step(s + 1, (s + 1) & 1);
          }
          const int buf = s & 1;
          if (s < NSN) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) { num[nt] = mfma16(fa[buf][e], fb[buf][nt][e], num[nt]); PMF_QPAD(num[nt]); }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) { den[nt] = mfma16(fa[buf][e], fb[buf][nt][e], den[nt]); PMF_QPAD(den[nt]); }
          }
#ifndef PMF_QUAD_PAD
          if (s + 1 < NSA) {
#pragma unroll
            for (int g = 0; g < NT + 1; ++g) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
              __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            }
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * NT, 0);
#endif
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sX[(wv * NT + nt) * 64 + lane] = num[nt];
#ifdef PMF_STAMPS
        PMF_STAMP(ts3);
        acc_a += ts3 - ts0;
#endif
      } else {
        // ================= B slot: the W rule and P += W_b^T V_b over this wave's panels =================
        // The head of the B slot is a chain of latencies (exchange reads, the division) beside the other quad's dense MFMA
        // stream on this SIMD: at equal priority its VALU instructions get one issue slot per MFMA of the partner
        // (2.9-3.2 k cycles for what takes 0.65 k alone, in-kernel stamps); at the higher priority they issue when ready.
#ifndef PMF_QUAD_NOPRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        const bool more = (b + 1 < nb);
        f32x4 x[4][NT];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) x[r][nt] = sX[((4 * qd + r) * NT + nt) * 64 + lane];
        // the same sum in every wave of the quad: (x0 + x1) + (x2 + x3)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) num[nt] = (x[0][nt] + x[1][nt]) + (x[2][nt] + x[3][nt]);
        PMF_STAMP(ts1);
        f32x4 wn[NT];
        float* wdst = W + (size_t)blk * (16 * KP) + (4 * kq) * KP + NT * i;
        float tnum[NT][4], dd[NT][4], rr[NT][4], qq[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float w0 = wold[nt][j];
            if (MODE == FUSED_BNMF) {                              // bnmf.py:87-90
              tnum[nt][j] = num[nt][j] + (3.0f * lamb) * (w0 * w0);
              dd[nt][j] = ((den[nt][j] + (2.0f * lamb) * (w0 * w0 * w0)) + lamb * w0) + PMF_EPS_DEN;
            } else if (MODE == FUSED_RNMF) {                       // rnmf.py:109-115
              const float xx = num[nt][j];
              tnum[nt][j] = fabsf(xx) - xx;
              dd[nt][j] = 2.0f * den[nt][j];
            } else {
              tnum[nt][j] = w0 * num[nt][j];                       // nmf.py:131 (multiply first)
              dd[nt][j] = den[nt][j] + PMF_EPS_DEN;
            }
          }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) rr[nt][j] = __builtin_amdgcn_rcpf(dd[nt][j]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) qq[nt][j] = tnum[nt][j] * rr[nt][j];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) tnum[nt][j] = fmaf(-dd[nt][j], qq[nt][j], tnum[nt][j]);   // residual
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float w = fmaf(tnum[nt][j], rr[nt][j], qq[nt][j]);        // pmf_div (nmf.py:132)
            if (MODE == FUSED_BNMF) w = wold[nt][j] * w;
            if (MODE == FUSED_RNMF) w = dd[nt][j] != 0.f ? wold[nt][j] * w : 0.f;
            wn[nt][j] = w;
            if (h == j) wrow_store(&wdst[j * KP + nt], w);            // a quarter of the rows per wave
          }
#ifndef PMF_QUAD_NOPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        PMF_STAMP(ts2);
        // S += W_b^T W_b, this wave's quarter: MFMA step h contracts rows {4q + h}; at the head of the dense stream
        {
          float ws[NT];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) ws[nt] = h == 0 ? wn[nt][0] : h == 1 ? wn[nt][1] : h == 2 ? wn[nt][2] : wn[nt][3];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt <= nt; ++mt) S[mt][nt] = mfma16(ws[mt], ws[nt], S[mt][nt]);
        }
        // ---------------- phase B: P += W_b^T V_b ----------------
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
            for (int nt = 0; nt < 4; ++nt) { P[mt][4 * p + nt] = mfma16(wn[mt][j], bf[buf][nt], P[mt][4 * p + nt]); PMF_QPAD(P[mt][4 * p + nt]); }
#ifndef PMF_QUAD_PAD
          if (s + 1 < NSN) {
            __builtin_amdgcn_sched_group_barrier(0x008, NT >= 2 ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#endif
          __builtin_amdgcn_sched_barrier(0);
          if (j == 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of panel p has returned: refill it
            if (more) {
#pragma unroll
              for (int q = 0; q < 4; ++q) issue_v(blk + 1, p, q);
            }
          }
        }
        PMF_STAMP(ts3);
#ifdef PMF_STAMPS
        acc_x += ts1 - ts0; acc_epi += ts2 - ts1; acc_b += ts3 - ts2;
#endif
      }
    }
    PMF_STAMP(ts3);
    quad_barrier();
    PMF_STAMP(ts4);
#ifdef PMF_STAMPS
    acc_bar += ts4 - ts3;
#endif
  }
#ifdef PMF_STAMPS
  PMF_STAMP(tk2);
#endif
  // ---- quad 1's P tiles and all eight S quarters through LDS; quad 0 writes the P tiles from registers, quad 1 the S tiles ----
  // (the loop's last barrier has passed: no wave reads H, G or a tile any more)
  f32x4* exP = reinterpret_cast<f32x4*>(smem);        // [4 waves][NT * NTP tiles][64]
  f32x4* exS = exP + (size_t)4 * NT * NTP * 64;       // [8 waves][NS tiles][64]
  static_assert(((size_t)4 * NT * NTP + 8 * NS) * 64 * 16 <= quad_smem_bytes<NT, NPANEL>(), "exchange fits");
  if (qd == 1) {
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTP; ++nt) exP[((h * NT + mt) * NTP + nt) * 64 + lane] = P[mt][nt];
  }
#pragma unroll
  for (int mt = 0; mt < NT; ++mt)
#pragma unroll
    for (int nt = mt; nt < NT; ++nt) exS[(wv * NS + mt * NT - (mt * (mt - 1)) / 2 + (nt - mt)) * 64 + lane] = S[mt][nt];
  __syncthreads();
  constexpr int NTPT = 4 * NPT, NTUT = NT * NTPT + NS;
  f32x4* out = reinterpret_cast<f32x4*>(slab) + (size_t)blockIdx.x * NTUT * 64 + lane;
  if (qd == 0) {
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTP; ++nt)
        slab_store16(&out[(mt * NTPT + h * NTP + nt) * 64], P[mt][nt] + exP[((h * NT + mt) * NTP + nt) * 64 + lane]);
  } else if (h < NS) {
    f32x4 s = exS[(0 * NS + h) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) s += exS[(w * NS + h) * 64 + lane];
    slab_store16(&out[(NT * NTPT + h) * 64], s);
  }
#ifdef PMF_STAMPS
  PMF_STAMP(tk3);
  if (dbg && lane == 0) {
    unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wv) * 8;
    d[0] = acc_a; d[1] = acc_x; d[2] = acc_epi; d[3] = acc_b; d[4] = acc_bar; d[5] = (unsigned long long)nb;
    d[6] = tk1 - tk0; d[7] = tk3 - tk2; (void)acc_f1; (void)acc_f2;
  }
#endif
}

#ifndef PMF_FUSED_KERNEL_ONLY
// Shapes served: 32 bases (NT = 2) on 512 padded columns.
static inline bool fused_shape_quad(int NT, int np) { return NT == 2 && np == 512; }

template <int NT, int NPANEL, int MODE>
static int launch_quad_t(hipStream_t s, const float* V, float* W, const float* H, const float* G, int64_t mp, int wgs,
                         float lamb, float* slab, const FusedCtl& ctl, int ngp) {
  const int nblk = (int)(mp / 16), nq = wgs * 2;     // quads
  const int blk_per = nblk / nq, blk_extra = nblk % nq;
  const size_t smem = quad_smem_bytes<NT, NPANEL>();
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nmf_quad<NT, NPANEL, MODE>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return PMF_EHIP;
    attr_done = true;
  }
  hipLaunchKernelGGL((k_nmf_quad<NT, NPANEL, MODE>), dim3(wgs), dim3(512), smem, s, V, W, H, G, blk_per, blk_extra, lamb,
                     slab, ctl, ngp);
  return PMF_OK;
}
#endif
