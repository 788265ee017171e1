// pmf_small.h -- the k x k / k x n sized kernels around the two big contractions.
#pragma once
#include <hip/hip_cooperative_groups.h>
#include "pmf_dev.h"
#include "pmf_ipc.h"

// NMF H step (pymf/nmf.py:122-126): H <- (H * P) / (S H + 1e-9), P = W^T V, S = W^T W.
// PS: [KP][np + KP] (P | S).  One block per 16 columns of H; in place.
// mode 1: BNMF rule (bnmf.py:79-82) H *= (P + 3 l H^2) / (S H + 2 l H^3 + l H + 1e-9).
// mode 2: RNMF rule (rnmf.py:100-105) with P = W^T (S - data): H *= (|P| - P) / (2 (W^T W) H),
//         no epsilon; kvalid / nvalid mask the zero padding (0/0 there).
// mode 3: SNMF rule (snmf.py:72-91) for num_bases > 128 (k_snmf_h_mfma serves the rest):
//         H *= sqrt((pos(P) + neg(S) H) / (neg(P) + pos(S) H + 1e-9)).
__global__ __launch_bounds__(256) void k_nmf_h(float* __restrict__ H, int64_t ldh, int np, int KP,
                                               const float* __restrict__ PS, int bnmf, float lamb,
                                               int kvalid, int nvalid) {
  extern __shared__ __attribute__((aligned(16))) float hs[];   // [KP][16]
  const int tid = threadIdx.x;
  const int c = tid & 15;
  const int col = blockIdx.x * 16 + c;
  const int64_t ldp = (int64_t)np + KP;
  for (int kk = tid >> 4; kk < KP; kk += 16) hs[kk * 16 + c] = H[(int64_t)kk * ldh + col];
  __syncthreads();
  for (int kk = tid >> 4; kk < KP; kk += 16) {
    const float* srow = PS + (int64_t)kk * ldp + np;
    const float h = hs[kk * 16 + c];
    const float p = PS[(int64_t)kk * ldp + col];
    if (bnmf == 3) {
      float ap = 0.f, an = 0.f;
      for (int j = 0; j < KP; ++j) {
        const float ww = srow[j], hj = hs[j * 16 + c];
        ap = fmaf((fabsf(ww) + ww) * 0.5f, hj, ap);              // snmf.py:73-74
        an = fmaf((fabsf(ww) - ww) * 0.5f, hj, an);              // snmf.py:76-77
      }
      const float h1 = (fabsf(p) + p) * 0.5f + an;
      const float h2 = (fabsf(p) - p) * 0.5f + ap + PMF_EPS_DEN;
      H[(int64_t)kk * ldh + col] = h * sqrtf(h1 / h2);
      continue;
    }
    float den = 0.f;
    for (int j = 0; j < KP; ++j) den = fmaf(srow[j], hs[j * 16 + c], den);
    if (bnmf == 2) {
      const float r = h * ((fabsf(p) - p) / (2.0f * den));
      H[(int64_t)kk * ldh + col] = (kk < kvalid && col < nvalid) ? r : 0.f;
    } else if (bnmf) {
      const float h1 = p + (3.0f * lamb) * (h * h);
      const float h2 = ((den + (2.0f * lamb) * (h * h * h)) + lamb * h) + PMF_EPS_DEN;
      H[(int64_t)kk * ldh + col] = h * (h1 / h2);
    } else {
      H[(int64_t)kk * ldh + col] = (h * p) / (den + PMF_EPS_DEN);
    }
  }
}

// NMF H step AND the Gram matrix of the new H in ONE launch (MFMA):
//   H <- (H * P) / (S H + 1e-9)   (pymf/nmf.py:122-126),  then  G = H H^T  (operand of the next
//   update_w, nmf.py:130 reassociated).
// Both are k x k x n sized, but on ONE CU their 2 * KP * KP * np / 1024 MFMAs are the whole
// duration (8 k cycles each at cfg4), so the columns are spread over workgroups: workgroup g takes
// the 64-column panels g, g + grid, ...; the H step of a panel needs nothing from other panels, the
// Gram matrix is a sum over panels.  Each workgroup keeps its partial G in registers across its
// panels, writes it to Gpart[g], and the LAST workgroup to arrive (atomic ticket) adds the partials
// in fixed order -- deterministic -- and writes G (float32 + float64) and the trace terms.
// Wave w of 16 owns tiles (mt, ct) = (q / 4, q % 4), q = w, w + 16, ... of the panel for the H step
// (the new values go to a second LDS image, the old one stays the B operand of the other waves)
// and tiles q = w, w + 16, ... of G.  LDS rows are padded by 4 floats: fragment reads are
// conflict-free b128.
//   tout (optional): the two data-dependent terms of the trace identity for the residual,
//   tout[0] = <P, H_new>,  tout[1] = <S H_new, H_new> = <S, G>  (float64 sums).
template <int NT>
constexpr size_t hgram_smem_bytes() { return (size_t)(16 * NT * (16 * NT + 4) + 2 * 16 * NT * 68) * sizeof(float); }

template <int NT, bool BNMF, bool FOLD>
__global__ __launch_bounds__(1024) void k_nmf_h_gram(float* __restrict__ H, int np,
                                                     float* __restrict__ PS,
                                                     float* __restrict__ Gf, double* __restrict__ Gd,
                                                     float lamb, double* __restrict__ tout,
                                                     float* Gpart, double* t1part, unsigned* ticket,
                                                     const int* __restrict__ stop, int final_sum,
                                                     IpcPeers pr, unsigned seq, int nflags, int* __restrict__ ipc_err,
                                                     unsigned long long wait_ticks, unsigned long long* __restrict__ waitstat) {
  // FOLD (round 5, the folded exchange; pr.nranks > 1): PS is NOT yet summed over the ranks -- every rank's k_reduce_slabs_tiles has
  // pushed its partial into slot [seq & 1] of this rank's receive area (nflags tiles, one flag each).  The prologue waits for
  // the flags of all ranks, forms S and its P panels as the sums of the N partials IN RANK ORDER (the bits of
  // k_ipc_allreduce and of the host transport) while loading them, and writes the sums to PS (S: workgroup 0, P: the owner
  // of the panel) for whoever reads (W^T V | W^T W) later.
  __shared__ double red[2][16];
  __shared__ unsigned s_last;
  __shared__ int s_ipc_ok;
  // (a template parameter since round 6: as a run-time branch on pr.nranks the eight mapped areas and the fold's addressing stayed
  // live beside the one-rank path and the 128-base instantiations spilled 88 / 120 bytes per lane at their 128-register budget)
  constexpr bool fold = FOLD;
  constexpr int KP = 16 * NT, LDS_S = KP + 4, LDS_H = 68;
  constexpr int HT = NT * 4;                  // H-step tiles of a panel
  constexpr int HTW = (HT + 15) / 16;         // ... per wave
  constexpr int GT = NT * NT;                 // G tiles
  constexpr int GTW = (GT + 15) / 16;
  constexpr int SQ = (KP * (KP / 4) + 1023) / 1024;   // 16-byte pieces of S / of an H panel per thread
  constexpr int HQ = (KP * 16 + 1023) / 1024;
  const int64_t ldp = (int64_t)np + KP;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* ss = sm;                    // [KP][KP+4]   S = W^T W
  float* hs = ss + KP * LDS_S;       // [KP][68]     old H panel
  float* hn = hs + KP * LDS_H;       // [KP][68]     new H panel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int npanel = np >> 6;
  // Everything the first panel needs is requested in ONE round trip -- S, the H panel, this wave's P values
  // (accumulator layout) and the stop flag -- and only then written to LDS: the kernel is a chain of memory
  // latencies (a k x n sized problem on a handful of CUs), three of them in a row before this order.
  f32x4 sreg[SQ], hreg[HQ];
  float pv[HTW][4];
  auto load_hpanel = [&](int p) {
    const int c0 = 64 * p;
#pragma unroll
    for (int u = 0; u < HQ; ++u) {
      const int q = tid + 1024 * u, r = q >> 4, c4 = q & 15;
      if (q < KP * 16) hreg[u] = *reinterpret_cast<const f32x4*>(H + (int64_t)r * np + c0 + 4 * c4);
    }
  };
  auto load_pv_fold = [&](int p) {                  // this wave's P values: sums of the ranks' partials, written back to PS
    // rank by rank (the order of ipc_sum1: 0 + p_0 + p_1 + ...), 32-bit element offsets from one scalar base per rank (the
    // payload is at most PMF_IPC_MAX_BYTES) and an opaque zero as in load_panel below: eight 64-bit row pointers per rank,
    // hoisted out of the panel loop, were the scratch of the 128-base instantiations
    const int c0 = 64 * p;
    int z = 0;
    asm volatile("" : "+v"(z));
    const char* base = pr.area[pr.me] + (size_t)(seq & 1u) * pr.nranks * PMF_IPC_MAX_BYTES;
#pragma unroll
    for (int h = 0; h < HTW; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) pv[h][r] = 0.f;
    for (int rk = 0; rk < pr.nranks; ++rk) {
      const float* src = reinterpret_cast<const float*>(base + (size_t)rk * PMF_IPC_MAX_BYTES);
#pragma unroll
      for (int h = 0; h < HTW; ++h) {
        const int q = wv + 16 * h, mt = q >> 2, ct = q & 3;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (q < HT) pv[h][r] += __builtin_nontemporal_load(src + ((16 * mt + 4 * kq + r + z) * (int)ldp + c0 + 16 * ct + i));
      }
    }
#pragma unroll
    for (int h = 0; h < HTW; ++h) {
      const int q = wv + 16 * h, mt = q >> 2, ct = q & 3;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (q < HT) PS[(16 * mt + 4 * kq + r + z) * (int)ldp + c0 + 16 * ct + i] = pv[h][r];
    }
  };
  auto load_panel = [&](int p) {
    const int c0 = 64 * p;
    load_hpanel(p);
    if (fold) { load_pv_fold(p); return; }
    // (an opaque zero in the row index: otherwise the 2 x 4 row pointers of this wave's P values are hoisted out of the
    // panel loop as eight 64-bit registers that live through the whole kernel -- the scratch of the 128-base instantiations)
    int z = 0;
    asm volatile("" : "+v"(z));
#pragma unroll
    for (int h = 0; h < HTW; ++h) {
      const int q = wv + 16 * h, mt = q >> 2, ct = q & 3;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        pv[h][r] = q < HT ? PS[(int64_t)(16 * mt + 4 * kq + r + z) * ldp + c0 + 16 * ct + i] : 0.f;
    }
  };
  if (fold) {
    if (stop != nullptr && *stop != 0) return;      // (before the wait: a stopped chunk pushed nothing, on any rank)
    if (blockIdx.x < npanel) load_hpanel(blockIdx.x);                 // H does not depend on the exchange: in flight during the wait
    const unsigned long long t0 = wall_clock64();
    if (!ipc_wait_all(pr, seq, nflags, wait_ticks, &s_ipc_ok)) { if (tid == 0) atomicExch(ipc_err, 1); return; }
    if (waitstat != nullptr && blockIdx.x == 0 && tid == 0) {         // what the exchange cost this iteration: the wait for the slowest peer
      atomicAdd(waitstat, wall_clock64() - t0);
      atomicAdd(waitstat + 1, 1ull);
    }
#pragma unroll
    for (int u = 0; u < SQ; ++u) {
      const int q = tid + 1024 * u, r = q / (KP / 4), c4 = q % (KP / 4);
      if (q < KP * (KP / 4)) {
        const int64_t e = (int64_t)r * ldp + np + 4 * c4;
        // (straight to LDS: nothing to overlap with after the wait, and four more live 16-byte registers beside the fold's
        // addressing spilled at 128 bases)
        const f32x4 sv = ipc_sum4(pr, seq, e);
        if (blockIdx.x == 0) *reinterpret_cast<f32x4*>(PS + e) = sv;
        *reinterpret_cast<f32x4*>(ss + r * LDS_S + 4 * c4) = sv;
      }
    }
    if (blockIdx.x < npanel) load_pv_fold(blockIdx.x);
  } else {
#pragma unroll
    for (int u = 0; u < SQ; ++u) {
      const int q = tid + 1024 * u, r = q / (KP / 4), c4 = q % (KP / 4);
      if (q < KP * (KP / 4)) sreg[u] = *reinterpret_cast<const f32x4*>(PS + (int64_t)r * ldp + np + 4 * c4);
    }
  }
  if (!fold && blockIdx.x < npanel) load_panel(blockIdx.x);
  if (stop != nullptr && *stop != 0) return;
  if (!fold) {
#pragma unroll
    for (int u = 0; u < SQ; ++u) {
      const int q = tid + 1024 * u, r = q / (KP / 4), c4 = q % (KP / 4);
      if (q < KP * (KP / 4)) *reinterpret_cast<f32x4*>(ss + r * LDS_S + 4 * c4) = sreg[u];
    }
  }
  // accumulator chains per G tile: four where a wave owns one tile (independent MFMAs back to back); at
  // num_bases > 64 a wave owns four tiles -- already four independent chains, and 64 accumulator registers
  // in four chains each spilled (128-register budget at 16 waves per workgroup)
  constexpr int NCH = GTW >= 4 ? 1 : 4;
  f32x4 ge[GTW][NCH];
#pragma unroll
  for (int g = 0; g < GTW; ++g)
#pragma unroll
    for (int e = 0; e < NCH; ++e) ge[g][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto gsum = [&](int g) -> f32x4 {
    if (NCH == 1) return ge[g][0];
    return (ge[g][0] + ge[g][NCH > 1 ? 1 : 0]) + (ge[g][NCH > 2 ? 2 : 0] + ge[g][NCH > 3 ? 3 : 0]);
  };
  double t1 = 0.0;

  for (int p = blockIdx.x; p < npanel; p += gridDim.x) {
    const int c0 = 64 * p;
    if (FOLD && p != blockIdx.x) break;               // (the folded payload is <= 256 KiB: np / 64 <= 63 panels, one per workgroup -- the host checks)
    if (!FOLD && p != blockIdx.x) {                   // (grids of fewer workgroups than panels: np > 4096)
      __syncthreads();                                // the previous panel's images are free
      load_panel(p);
    }
#pragma unroll
    for (int u = 0; u < HQ; ++u) {
      const int q = tid + 1024 * u, r = q >> 4, c4 = q & 15;
      if (q < KP * 16) *reinterpret_cast<f32x4*>(hs + r * LDS_H + 4 * c4) = hreg[u];
    }
    __syncthreads();
    // ---- H step ----
#pragma unroll
    for (int h = 0; h < HTW; ++h) {
      const int q = wv + 16 * h;
      if (q >= HT) break;
      const int mt = q >> 2, ct = q & 3;
      f32x4 den[4];                                   // 4 independent chains, one per element of a 16-byte group
#pragma unroll
      for (int e = 0; e < 4; ++e) den[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      // (at 128 bases the fully unrolled loop hoists all 8 x (4 + 4) LDS reads above the MFMAs: 68-92 bytes of scratch at
      // the 128-register budget of a 16-wave workgroup; four steps at a time stay in registers)
      constexpr int TUN = NT >= 8 ? 4 : NT;
#pragma unroll TUN
      for (int t = 0; t < NT; ++t) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(ss + (16 * mt + i) * LDS_S + 16 * t + 4 * kq);
#pragma unroll
        for (int e = 0; e < 4; ++e)                   // B[k = 16t+4kq+e][col = 16ct+i] = H[k][col]
          den[e] = mfma16(a4[e], hs[(16 * t + 4 * kq + e) * LDS_H + 16 * ct + i], den[e]);
      }
      const f32x4 dsum = (den[0] + den[1]) + (den[2] + den[3]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kk = 16 * mt + 4 * kq + r, col = 16 * ct + i;
        const float hv = hs[kk * LDS_H + col];
        float hnew;
        if (BNMF) {                                             // bnmf.py:79-82, H *= H1 / H2
          const float h1 = pv[h][r] + (3.0f * lamb) * (hv * hv);
          const float h2 = ((dsum[r] + (2.0f * lamb) * (hv * hv * hv)) + lamb * hv) + PMF_EPS_DEN;
          hnew = hv * (h1 / h2);
        } else {
          hnew = (hv * pv[h][r]) / (dsum[r] + PMF_EPS_DEN);     // multiply, then divide (nmf.py:125-126)
        }
        H[(int64_t)kk * np + c0 + col] = hnew;
        hn[kk * LDS_H + col] = hnew;
        t1 = fma((double)pv[h][r], (double)hnew, t1);
      }
    }
    __syncthreads();
    // ---- partial G += H_p H_p^T ----
#pragma unroll
    for (int g = 0; g < GTW; ++g) {
      const int q = wv + 16 * g;
      if (q >= GT) break;
      const int mt = q / NT, nt = q % NT;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(hn + (16 * mt + i) * LDS_H + 16 * t + 4 * kq);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(hn + (16 * nt + i) * LDS_H + 16 * t + 4 * kq);
#pragma unroll
        for (int e = 0; e < 4; ++e) ge[g][e % NCH] = mfma16(a4[e], b4[e], ge[g][e % NCH]);
      }
    }
  }

  // ---- partials out, ticket, the last workgroup finishes ----
  float* mine = Gpart + (size_t)blockIdx.x * KP * KP;
#pragma unroll
  for (int g = 0; g < GTW; ++g) {
    const int q = wv + 16 * g;
    if (q >= GT) break;
    const int mt = q / NT, nt = q % NT;
    const f32x4 gs = gsum(g);
#pragma unroll
    for (int r = 0; r < 4; ++r) mine[(16 * mt + 4 * kq + r) * KP + 16 * nt + i] = gs[r];
  }
  // This workgroup's share of <P,H> and of <S,G> = <S, sum of the partial G> (float64): the pairs are
  // added in workgroup order by whoever finishes -- the last workgroup below, or, when the partials
  // are left for the fused kernel to add (final_sum == 0), k_conv_check / the host.  Same numbers,
  // same order either way.
  {
    double t2 = 0.0;
#pragma unroll
    for (int g = 0; g < GTW; ++g) {
      const int q = wv + 16 * g;
      if (q >= GT) break;
      const int mt = q / NT, nt = q % NT;
      const f32x4 gs = gsum(g);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        t2 = fma((double)ss[(16 * mt + 4 * kq + r) * LDS_S + 16 * nt + i], (double)gs[r], t2);
    }
    t1 = wave_sum_f64(t1);
    t2 = wave_sum_f64(t2);
    if (lane == 0) { red[0][wv] = t1; red[1][wv] = t2; }
    __syncthreads();
    if (tid == 0) {
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int w = 0; w < 16; ++w) { a1 += red[0][w]; a2 += red[1][w]; }
      t1part[2 * blockIdx.x] = a1;
      t1part[2 * blockIdx.x + 1] = a2;
    }
  }
  if (!final_sum) return;
  // partials visible device-wide before the ticket: every storing wave drains its stores, the workgroup meets, ONE agent-scope
  // release (the write-back of this XCD's L2) in front of the ticket -- not a fence in each of the sixteen waves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    s_last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  // acquire: drop whatever this CU's L1 holds, then the partials are read with plain 16-byte loads
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const int nwg = gridDim.x;
  for (int q4 = tid; q4 < KP * KP / 4; q4 += 1024) {
    f32x4 g = {0.f, 0.f, 0.f, 0.f};                   // fixed order over workgroups
    const f32x4* src = reinterpret_cast<const f32x4*>(Gpart) + q4;
    for (int w = 0; w < nwg; ++w) g += src[(size_t)w * (KP * KP / 4)];
    *reinterpret_cast<f32x4*>(Gf + 4 * q4) = g;
    if (Gd) {
#pragma unroll
      for (int e = 0; e < 4; ++e) Gd[4 * q4 + e] = (double)g[e];
    }
  }
  if (tout && tid == 0) {
    double a1 = 0.0, a2 = 0.0;
    for (int w = 0; w < nwg; ++w) { a1 += t1part[2 * w]; a2 += t1part[2 * w + 1]; }
    tout[0] = a1;
    tout[1] = a2;
  }
  if (tid == 0) *ticket = 0u;                         // ready for the next launch (stream order)
}

// SNMF H step (pymf/snmf.py:72-91) with XW = P^T (P = W^T V) and WW = S = W^T W:
//   H1 = pos(XW)^T + (H^T neg(WW))^T,  H2 = neg(XW)^T + (H^T pos(WW))^T + 1e-9,
//   H *= sqrt(H1 / H2).
// On MFMA, one workgroup per 64-column panel (columns are independent): wave w owns
// tiles (mt, ct) = (q / 4, q % 4), q = w, w + 16, ...; S is split into its positive and negative
// parts in registers as the A fragments are read, so pos(WW) H and neg(WW) H are two accumulator
// chains over the same operands.  LDS layout and staging as in k_nmf_h_gram.
// CT = column tiles per workgroup: 4 (64-column panels) or 1 (16-column panels, for narrow H: at n = 128
// two 64-column workgroups leave the step latency-bound on two CUs -- 14.7 us at k = 128 -- eight don't).
template <int NT, int CT>
constexpr size_t snmf_h_smem_bytes() { return (size_t)(16 * NT * (16 * NT + 4) + 16 * NT * (16 * CT + 4)) * sizeof(float); }

template <int NT, int CT>
__global__ __launch_bounds__(1024) void k_snmf_h_mfma(float* __restrict__ H, int np,
                                                      const float* __restrict__ PS,
                                                      const int* __restrict__ stop) {
  if (stop != nullptr && *stop != 0) return;
  constexpr int KP = 16 * NT, LDS_S = KP + 4, LDS_H = 16 * CT + 4;
  constexpr int HT = NT * CT, HTW = (HT + 15) / 16;
  const int64_t ldp = (int64_t)np + KP;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* ss = sm;                    // [KP][KP+4]   WW = W^T W
  float* hs = ss + KP * LDS_S;       // [KP][16 CT + 4]   H panel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int c0 = 16 * CT * blockIdx.x;
  float pv[HTW][4];
#pragma unroll
  for (int h = 0; h < HTW; ++h) {
    const int q = wv + 16 * h, mt = q / CT, ct = q % CT;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      pv[h][r] = q < HT ? PS[(int64_t)(16 * mt + 4 * kq + r) * ldp + c0 + 16 * ct + i] : 0.f;
  }
  for (int q = tid; q < KP * (KP / 4); q += 1024) {
    const int r = q / (KP / 4), c4 = q % (KP / 4);
    *reinterpret_cast<f32x4*>(ss + r * LDS_S + 4 * c4) =
        *reinterpret_cast<const f32x4*>(PS + (int64_t)r * ldp + np + 4 * c4);
  }
  for (int q = tid; q < KP * 4 * CT; q += 1024) {
    const int r = q / (4 * CT), c4 = q % (4 * CT);
    *reinterpret_cast<f32x4*>(hs + r * LDS_H + 4 * c4) =
        *reinterpret_cast<const f32x4*>(H + (int64_t)r * np + c0 + 4 * c4);
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < HTW; ++h) {
    const int q = wv + 16 * h;
    if (q >= HT) break;
    const int mt = q / CT, ct = q % CT;
    f32x4 accp[2], accn[2];            // 2 chains each
#pragma unroll
    for (int e = 0; e < 2; ++e) { accp[e] = f32x4{0.f, 0.f, 0.f, 0.f}; accn[e] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(ss + (16 * mt + i) * LDS_S + 16 * t + 4 * kq);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ww = a4[e];                                  // WW[kk][j] = WW[j][kk]
        const float wp = (fabsf(ww) + ww) * 0.5f;                // snmf.py:73-74
        const float wn = (fabsf(ww) - ww) * 0.5f;                // snmf.py:76-77
        const float hj = hs[(16 * t + 4 * kq + e) * LDS_H + 16 * ct + i];
        accp[e & 1] = mfma16(wp, hj, accp[e & 1]);
        accn[e & 1] = mfma16(wn, hj, accn[e & 1]);
      }
    }
    const f32x4 a2 = accp[0] + accp[1], a1 = accn[0] + accn[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = 16 * mt + 4 * kq + r, col = 16 * ct + i;
      const float xw = pv[h][r];
      const float h1 = (fabsf(xw) + xw) * 0.5f + a1[r];
      const float h2 = (fabsf(xw) - xw) * 0.5f + a2[r] + PMF_EPS_DEN;
      H[(int64_t)kk * np + c0 + col] = hs[kk * LDS_H + col] * sqrtf(h1 / h2);
    }
  }
}

// inv(G), G = H H^T, for matrix orders beyond k_inverse_spd_mfma (pmf_inv.h; num_bases > 128): in-place
// Gauss-Jordan without pivoting (on an SPD matrix every pivot is a positive Schur complement and the
// elimination is as stable as Cholesky), float64, the matrix in global memory (L2) and spread over a
// cooperative grid.  Step p maps every entry by
//   a_rc - a_rp a_pc / a_pp  (r, c != p),   a_pc / a_pp  (row p),   -a_rp / a_pp  (column p),   1 / a_pp
// from the state BEFORE the step, so the matrix ping-pongs between two buffers (A0 holds G on entry)
// and ONE grid barrier per pivot is enough.  Identity padding (rows/cols >= k) is left alone: those
// pivots are 1 with zero row and column.  The result ends in Ginv64.
__global__ __launch_bounds__(1024) void k_inverse_spd_big(double* A0, double* A1, int KP, int k,
                                                          double* __restrict__ Ginv64,
                                                          const int* __restrict__ stop,
                                                          int* __restrict__ singular) {
  if (stop != nullptr && *stop != 0) return;          // uniform over the grid: nobody reaches a barrier
  cooperative_groups::grid_group grid = cooperative_groups::this_grid();
  const int64_t gtid = (int64_t)blockIdx.x * 1024 + threadIdx.x, gsize = (int64_t)gridDim.x * 1024;
  const int64_t E = (int64_t)KP * KP;
  double* A = A0;
  double* An = A1;
  for (int p = 0; p < k; ++p) {
    const double app = A[(int64_t)p * KP + p];
    const double inv = 1.0 / app;
    for (int64_t q = gtid; q < E; q += gsize) {
      const int r = (int)(q / KP), c = (int)(q % KP);
      const double arc = A[q];
      double v;
      if (r == p) v = (c == p) ? inv : arc * inv;
      else if (c == p) v = -arc * inv;
      else v = fma(-A[(int64_t)r * KP + p] * inv, A[(int64_t)p * KP + c], arc);
      An[q] = v;
    }
    grid.sync();
    { double* x = A; A = An; An = x; }
  }
  bool bad = false;                                   // a zero pivot leaves inf / nan behind
  for (int64_t q = gtid; q < E; q += gsize) { const double v = A[q]; bad |= !(fabs(v) <= 1.7e308); Ginv64[q] = v; }
  if (singular != nullptr && bad) *singular = 1;
}

// out = a + b over count floats (count a multiple of 4): RNMF's S = D + V for pmf_rnmf_get_s_f32.
__global__ __launch_bounds__(256) void k_add_f32(const float* __restrict__ a, const float* __restrict__ b, int64_t count,
                                                 float* __restrict__ out) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; 4 * q < count; q += (int64_t)gridDim.x * 256)
    reinterpret_cast<f32x4*>(out)[q] = reinterpret_cast<const f32x4*>(a)[q] + reinterpret_cast<const f32x4*>(b)[q];
}

// dst += src (count a multiple of 4): the chunk sums of a product over very many columns (pmf_api.hip: PMF_WIDE_K).
__global__ __launch_bounds__(256) void k_acc_f32(float* __restrict__ dst, const float* __restrict__ src, int64_t count) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; 4 * q < count; q += (int64_t)gridDim.x * 256) {
    f32x4 d = reinterpret_cast<f32x4*>(dst)[q];
    d += reinterpret_cast<const f32x4*>(src)[q];
    reinterpret_cast<f32x4*>(dst)[q] = d;
  }
}

// dst -= src (count a multiple of 4).
__global__ __launch_bounds__(256) void k_sub_f32(float* __restrict__ dst, const float* __restrict__ src, int64_t count) {
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; 4 * q < count; q += (int64_t)gridDim.x * 256) {
    f32x4 d = reinterpret_cast<f32x4*>(dst)[q];
    d -= reinterpret_cast<const f32x4*>(src)[q];
    reinterpret_cast<f32x4*>(dst)[q] = d;
  }
}

// Den[r][j] = sum_i W[r][i] G[i][j] for a [rows][KP] W (KP <= 128): the small second product of the W rules where the
// first one (V H^T over very many columns) was formed in chunks and the rule is applied element by element (k_nmf_w_elem).
__global__ __launch_bounds__(256) void k_den_small(const float* __restrict__ W, const float* __restrict__ G, float* __restrict__ Den,
                                                   int64_t rows, int KP) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < rows * KP; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / KP;
    const int j = (int)(e % KP);
    float s = 0.f;
    for (int i = 0; i < KP; ++i) s = fmaf(W[r * KP + i], G[(int64_t)i * KP + j], s);
    Den[e] = s;
  }
}

// Per-block float64 partials of sum(X^2) over a padded [rows][ld] buffer (padding is zero).
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ X, int64_t count,
                                               double* __restrict__ part) {
  __shared__ double ws[4];
  double s = 0.0;
  const int64_t n4 = count >> 2;
  const f32x4* X4 = reinterpret_cast<const f32x4*>(X);
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
    const f32x4 v = X4[q];
    s += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
  }
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// Trace identity for the residual: with P = W^T V, S = W^T W of the CURRENT W,
//   ||V - W H||^2 = ||V||^2 - 2 <P, H> + <S H, H>.
// One block per 16 columns of H; part[2*b] = sum P.H, part[2*b+1] = sum (S H).H, float64.
// TH / TP: float (H, and P | S as the float32 roundings in (P | S)) or double (round 6, SNMF: the float64 H of pmf_inv.h; inside
// the Gram-space loop also P and S as the float64 products dPd / dSd -- the identity cancels ||V||^2 against <P,H>, and the
// float32 rounding of P alone cost 1e-4 of ferr on a fit at 7 % of ||V||).
template <typename TH, typename TP>
__global__ __launch_bounds__(256) void k_trace_terms(const TH* __restrict__ H, int64_t ldh, int np,
                                                     int KP, const TP* __restrict__ P, int64_t ldP,
                                                     const TP* __restrict__ S, int64_t ldS,
                                                     double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hs_raw[];   // [KP][16] of TH
  TH* hs = reinterpret_cast<TH*>(hs_raw);
  __shared__ double w1[4], w2[4];
  const int tid = threadIdx.x;
  const int c = tid & 15;
  const int col = blockIdx.x * 16 + c;
  for (int kk = tid >> 4; kk < KP; kk += 16) hs[kk * 16 + c] = H[(int64_t)kk * ldh + col];
  __syncthreads();
  double t1 = 0.0, t2 = 0.0;
  for (int kk = tid >> 4; kk < KP; kk += 16) {
    const TP* srow = S + (int64_t)kk * ldS;
    double sh = 0.0;
    for (int j = 0; j < KP; ++j) sh = fma((double)srow[j], (double)hs[j * 16 + c], sh);
    const double h = (double)hs[kk * 16 + c];
    t1 = fma((double)P[(int64_t)kk * ldP + col], h, t1);
    t2 = fma(sh, h, t2);
  }
  t1 = wave_sum_f64(t1);
  t2 = wave_sum_f64(t2);
  if ((tid & 63) == 0) { w1[tid >> 6] = t1; w2[tid >> 6] = t2; }
  __syncthreads();
  if (tid == 0) {
    part[2 * blockIdx.x] = w1[0] + w1[1] + w1[2] + w1[3];
    part[2 * blockIdx.x + 1] = w2[0] + w2[1] + w2[2] + w2[3];
  }
}

// out[0] = sum of part[0], part[2], ...; out[1] = sum of part[1], part[3], ... (n pairs)
__global__ void k_sum_pairs_f64(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double ws[2][4];
  double a = 0.0, b = 0.0;
  for (int q = threadIdx.x; q < n; q += 256) { a += part[2 * q]; b += part[2 * q + 1]; }
  a = wave_sum_f64(a);
  b = wave_sum_f64(b);
  if ((threadIdx.x & 63) == 0) { ws[0][threadIdx.x >> 6] = a; ws[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = ws[0][0] + ws[0][1] + ws[0][2] + ws[0][3];
    out[1] = ws[1][0] + ws[1][1] + ws[1][2] + ws[1][3];
  }
}

// sum of part[0..n) in float64, fixed order; out[0] = sum.
__global__ void k_sum_f64(const double* __restrict__ part, int n, double* __restrict__ out) {
  __shared__ double ws[4];
  double s = 0.0;
  // eight requests in flight, added in the order of the plain loop (one load per round trip took 26 us for 16 384 partials)
  for (int q = threadIdx.x; q < n; q += 8 * 256) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (q + 256 * u < n) ? part[q + 256 * u] : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ws[0] + ws[1] + ws[2] + ws[3];
}

// Synthetic U[0,1) fill of the logical rows x cols block of a padded [.,ld] buffer.
__global__ void k_fill_uniform(float* __restrict__ X, int64_t ld, int64_t rows, int64_t cols,
                               int64_t row0, int64_t cols_global, uint64_t seed) {
  // grid-stride: a launch of one thread per element wraps silently beyond 2^32 threads (a 36 Mi x 256 matrix has 9.7e9 elements)
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < rows * cols; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / cols, c = e % cols;
    X[r * ld + c] = u01_from(seed, (uint64_t)((row0 + r) * cols_global + c));
  }
}

// ---- host <-> device transport of V / W / H (pmf_set_*_f32 / _f64, pmf_get_*_f64) ----------------------------------
// The bytes cross PCIe as the host holds them -- ONE contiguous copy (56 GB/s from pageable memory on this part; a pitched
// hipMemcpy2DAsync reaches 17) -- and the zero padding to [.][dld] and the float64 -> float32 rounding happen here, at HBM
// speed, instead of in a single host thread (np.ascontiguousarray(float64 -> float32) of a 1 048 576 x 64 W: 0.2 s).
// dst [rows][dld] float32 (columns >= cols are written as zero), src [rows][cols] contiguous T.
template <typename T>
__global__ __launch_bounds__(256) void k_unpack_rows(const T* __restrict__ src, int64_t rows, int64_t cols, float* __restrict__ dst, int64_t dld) {
  const int64_t total = rows * dld;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / dld, c = e - r * dld;
    dst[e] = c < cols ? (float)src[r * cols + c] : 0.f;
  }
}
// dst [rows][cols] contiguous T, src [rows][sld] float32
template <typename T>
__global__ __launch_bounds__(256) void k_pack_rows(const float* __restrict__ src, int64_t sld, int64_t rows, int64_t cols, T* __restrict__ dst) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / cols, c = e - r * cols;
    dst[e] = (T)src[r * sld + c];
  }
}

// ---- streamed V (pmf_stream_*): accumulators that live across the tiles of one pass ----------
// acc[e] (+)= sum over slabs of slab[c][e]  (float64; first != 0 starts a new pass)
__global__ __launch_bounds__(256) void k_reduce_slabs_acc(const float* __restrict__ slab, int nslabs, int64_t E,
                                                          double* __restrict__ acc, int first) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  double s = first ? 0.0 : acc[e];
  for (int c = 0; c < nslabs; ++c) s += (double)slab[(int64_t)c * E + e];
  acc[e] = s;
}

__global__ void k_accum_f64(double* __restrict__ dst, const double* __restrict__ src, int first) {
  if (threadIdx.x == 0 && blockIdx.x == 0) dst[0] = (first ? 0.0 : dst[0]) + src[0];
}

__global__ __launch_bounds__(256) void k_f64_to_f32(const double* __restrict__ src, int64_t E, float* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < E) dst[e] = (float)src[e];
}

// Free-running pmf_factorize loop: iteration i's error and the reference's convergence test on the
// device (nmf.py:134-139,198-202), so the host does not have to read ferr back after every
// iteration.  tt = (<P,H>, <S,G>) from k_nmf_h_gram; ||V - W H||^2 = ||V||^2 - 2 tt[0] + tt[1].
// stop[0]: 0 running, 1 converged at iteration stop[1], 2 the identity cancels at iteration
// stop[1] (its error must be evaluated directly); once set, every later launch is a no-op.
__global__ void k_conv_check(const double* __restrict__ tt, int ntt, double vnorm2, double eps, double nsamp,
                             int i, double* __restrict__ ferr, int* __restrict__ stop) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || stop[0] != 0) return;
  double t0 = tt[0], t1 = tt[1];
  for (int q = 1; q < ntt; ++q) { t0 += tt[2 * q]; t1 += tt[2 * q + 1]; }   // ntt > 1: per-workgroup pairs
  const double e2 = vnorm2 - 2.0 * t0 + t1;
  if (!(e2 > 1e-3 * vnorm2)) { stop[1] = i; stop[0] = 2; return; }
  const double f = sqrt(e2);
  ferr[i] = f;
  if (i > 1 && fabs(f - ferr[i - 1]) / nsamp < eps) { stop[1] = i; stop[0] = 1; }
}

// G = sum of the per-workgroup partial Gram matrices k_nmf_h_gram leaves when the fused kernel is not
// the next consumer (fixed order).
__global__ __launch_bounds__(256) void k_sum_gparts(const float* __restrict__ Gpart, int nparts, int E,
                                                    float* __restrict__ G) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= E) return;
  float g = 0.f;
  for (int w = 0; w < nparts; ++w) g += Gpart[(size_t)w * E + q];
  G[q] = g;
}
