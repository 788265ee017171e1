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
// Accumulators: P = NT x 4*NPANEL tiles and S = NT x NT tiles of 16x16 (4 VGPRs
// each) stay in registers for the wave's whole row range; one wave per SIMD
// (__launch_bounds__(256, 1)) so the 512-entry unified VGPR/AGPR file holds them.
#pragma once
#include "pmf_dev.h"
#include "../../include/pymf_hip.h"

#define PMF_GLDS16(gsrc, ldst)                                                            \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc), \
                                   (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt range");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NT, int NPANEL>
constexpr size_t fused_smem_bytes() {
  return (size_t)64 * (NPANEL * 16 * NT + 16 * NT + 4 * 16 * NPANEL + 4 * 16) * sizeof(float);
}

template <int NT, int NPANEL>
__global__ __launch_bounds__(256, 1) void k_nmf_fused(const float* __restrict__ V,
                                                       float* __restrict__ W,
                                                       const float* __restrict__ H,
                                                       const float* __restrict__ G, int64_t mp,
                                                       float* __restrict__ slab) {
  constexpr int KP = 16 * NT;
  constexpr int NP = 64 * NPANEL;
  constexpr int NTP = 4 * NPANEL;   // column tiles of P
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sH = smem;                             // [NPANEL][KP][64]   swizzled rows
  float* sG = sH + NPANEL * KP * 64;            // [KP][64]
  float* sVall = sG + KP * 64;                  // 4 waves x [NPANEL][16][64]
  float* sWall = sVall + 4 * NPANEL * 16 * 64;  // 4 waves x [16][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (SGPR)
  const int i = lane & 15, kq = lane >> 4;
  float* sV = sVall + wv * (NPANEL * 1024);
  float* sW = sWall + wv * 1024;

  // ---- H and G into LDS (whole workgroup, once) ----
  for (int q = tid; q < NPANEL * KP * 16; q += 256) {
    const int p = q / (KP * 16), rem = q % (KP * 16);
    const int row = rem >> 4, c = rem & 15;
    lds_write4(sH + p * (KP * 64), row, c,
               *reinterpret_cast<const f32x4*>(H + (int64_t)row * NP + 64 * p + 4 * c));
  }
  for (int q = tid; q < KP * 16; q += 256) {
    const int row = q >> 4, c = q & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (4 * c < KP) v = *reinterpret_cast<const f32x4*>(G + row * KP + 4 * c);
    lds_write4(sG, row, c, v);
  }
  __syncthreads();

  // ---- this wave's contiguous range of 16-row blocks ----
  const int64_t nblk = mp >> 4;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wv, nw = (int64_t)gridDim.x * 4;
  const int64_t per = nblk / nw, extra = nblk % nw;
  const int64_t b0 = gw * per + (gw < extra ? gw : extra);
  const int64_t nb = per + (gw < extra ? 1 : 0);

  // LDS-DMA geometry: one instruction = 4 rows x 256 B; lane L fills physical chunk
  // (L & 15) of row 4q + (L >> 4), so it fetches logical chunk (L & 15) ^ (row & 15).
  const int drow = lane >> 4;        // + 4q
  const int dchunk = lane & 15;

  auto issue_v_panel = [&](int64_t r0, int p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 4 * q + drow;
      const float* src = V + (r0 + row) * NP + 64 * p + 4 * (dchunk ^ (row & 15));
      PMF_GLDS16(src, sV + p * 1024 + q * 256);
    }
  };
  auto issue_w = [&](int64_t r0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 4 * q + drow;
      int c = dchunk ^ (row & 15);
      if (4 * c >= KP) c = 0;        // beyond the k columns: any valid address, never read
      const float* src = W + (r0 + row) * KP + 4 * c;
      PMF_GLDS16(src, sW + q * 256);
    }
  };

  f32x4 P[NT][NTP];
  f32x4 S[NT][NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NTP; ++nt) P[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) S[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  if (nb > 0) {
    issue_w(b0 * 16);
#pragma unroll
    for (int p = 0; p < NPANEL; ++p) issue_v_panel(b0 * 16, p);
  }

  for (int64_t b = 0; b < nb; ++b) {
    const int64_t r0 = (b0 + b) * 16;
    const bool more = (b + 1 < nb);

    // ---------------- phase A: Num = V_b H^T ----------------
    f32x4 num[NT], den[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      num[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      den[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int p = 0; p < NPANEL; ++p) {
      // panels p+1.. of this block (4 DMA each) may still be in flight
      if (p == 0) wait_vmcnt<4 * (NPANEL - 1)>();
      else if (p == 1) wait_vmcnt<(NPANEL > 1 ? 4 * (NPANEL - 2) : 0)>();
      else if (p == 2) wait_vmcnt<(NPANEL > 2 ? 4 * (NPANEL - 3) : 0)>();
      else wait_vmcnt<0>();
      const float* vp = sV + p * 1024;
      const float* hp = sH + p * (KP * 64);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int chunk = 4 * t + kq;
        const f32x4 a4 = lds_read4(vp, i, chunk);
        f32x4 b4[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b4[nt] = lds_read4(hp, 16 * nt + i, chunk);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) num[nt] = mfma16(a4[e], b4[nt][e], num[nt]);
      }
    }
    // ---------------- Den = W_b G ----------------
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int chunk = 4 * t + kq;
      const f32x4 a4 = lds_read4(sW, i, chunk);
      f32x4 b4[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b4[nt] = lds_read4(sG, 16 * nt + i, chunk);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) den[nt] = mfma16(a4[e], b4[nt][e], den[nt]);
    }
    // ---------------- epilogue: W_b <- (W_b * Num) / (Den + eps) ----------------
    f32x4 wn[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 4 * kq + j, col = 16 * nt + i;
        const float wold = sW[swz_off(row, col >> 2) + (col & 3)];
        const float w = (wold * num[nt][j]) / (den[nt][j] + PMF_EPS_DEN);
        wn[nt][j] = w;
        W[(r0 + row) * KP + col] = w;
      }
    // the W image is free again: prefetch the next block's rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (more) issue_w(r0 + 16);

    // ---------------- phase B: P += W_b^T V_b, S += W_b^T W_b ----------------
#pragma unroll
    for (int p = 0; p < NPANEL; ++p) {
      const float* vp = sV + p * 1024;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 4 * kq + j;
        float bf[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bf[nt] = vp[swz_off(row, 4 * nt + (i >> 2)) + (i & 3)];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
            P[mt][4 * p + nt] = mfma16(wn[mt][j], bf[nt], P[mt][4 * p + nt]);
      }
      // every read of panel p has returned: refill it for the next block
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (more) issue_v_panel(r0 + 16, p);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) S[mt][nt] = mfma16(wn[mt][j], wn[nt][j], S[mt][nt]);
  }

  // ---- sum the 4 waves' accumulators through LDS (tree), wave 0 writes the slab ----
  constexpr int NTILE = NT * (NTP + NT);
  __syncthreads();
  f32x4* ex = reinterpret_cast<f32x4*>(smem);   // two regions of NTILE*64 f32x4
  static_assert((size_t)2 * NTILE * 64 * 16 <= fused_smem_bytes<NT, NPANEL>(), "exchange fits");
  auto put = [&](int region) {
    f32x4* dst = ex + (size_t)region * NTILE * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NTP; ++nt) dst[(mt * (NTP + NT) + nt) * 64] = P[mt][nt];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) dst[(mt * (NTP + NT) + NTP + nt) * 64] = S[mt][nt];
    }
  };
  auto add = [&](int region) {
    const f32x4* src = ex + (size_t)region * NTILE * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NTP; ++nt) P[mt][nt] += src[(mt * (NTP + NT) + nt) * 64];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) S[mt][nt] += src[(mt * (NTP + NT) + NTP + nt) * 64];
    }
  };
  if (wv & 1) put(wv >> 1);
  __syncthreads();
  if (!(wv & 1)) add(wv >> 1);
  __syncthreads();
  if (wv == 2) put(0);
  __syncthreads();
  if (wv == 0) {
    add(0);
    const int64_t ldp = (int64_t)NP + KP;
    float* base = slab + (int64_t)blockIdx.x * KP * ldp;
#pragma unroll
    for (int mt = 0; mt < NT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* rowp = base + (int64_t)(16 * mt + 4 * kq + j) * ldp;
#pragma unroll
        for (int nt = 0; nt < NTP; ++nt) rowp[16 * nt + i] = P[mt][nt][j];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rowp[NP + 16 * nt + i] = S[mt][nt][j];
      }
  }
}

// ---- host-side dispatch -----------------------------------------------------------------
#ifndef PMF_FUSED_KERNEL_ONLY
static inline bool fused_shape_ok(int NT, int np) {
  const int npanel = np / 64;
  return (NT == 1 || NT == 2 || NT == 4) && npanel >= 1 && npanel <= 4 && npanel != 3 && np % 64 == 0;
}

// Workgroups to launch (one per CU), 0 when the shape is not covered by the fused kernel.
static inline int fused_grid_for(int NT, int np, int64_t mp) {
  if (!fused_shape_ok(NT, np)) return 0;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    cus = prop.multiProcessorCount;
  const int64_t nblk = mp / 16;
  int64_t wgs = (nblk + 3) / 4;
  if (wgs > cus) wgs = cus;
  return (int)wgs;
}

static inline const char* fused_kernel_name(int NT, int np) {
  static char buf[64];
  snprintf(buf, sizeof(buf), "k_nmf_fused<%d,%d>", NT, np / 64);
  return buf;
}

template <int NT, int NPANEL>
static int launch_fused_t(hipStream_t s, const float* V, float* W, const float* H, const float* G,
                          int64_t mp, int wgs, float* slab) {
  const size_t smem = fused_smem_bytes<NT, NPANEL>();
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nmf_fused<NT, NPANEL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return PMF_EHIP;
    attr_done = true;
  }
  hipLaunchKernelGGL((k_nmf_fused<NT, NPANEL>), dim3(wgs), dim3(256), smem, s, V, W, H, G, mp, slab);
  return PMF_OK;
}

static inline int launch_fused(hipStream_t s, int NT, int np, const float* V, float* W, const float* H,
                               const float* G, int64_t mp, int wgs, float* slab) {
  const int key = NT * 10 + np / 64;
  switch (key) {
    case 11: return launch_fused_t<1, 1>(s, V, W, H, G, mp, wgs, slab);
    case 12: return launch_fused_t<1, 2>(s, V, W, H, G, mp, wgs, slab);
    case 14: return launch_fused_t<1, 4>(s, V, W, H, G, mp, wgs, slab);
    case 21: return launch_fused_t<2, 1>(s, V, W, H, G, mp, wgs, slab);
    case 22: return launch_fused_t<2, 2>(s, V, W, H, G, mp, wgs, slab);
    case 24: return launch_fused_t<2, 4>(s, V, W, H, G, mp, wgs, slab);
    case 41: return launch_fused_t<4, 1>(s, V, W, H, G, mp, wgs, slab);
    case 42: return launch_fused_t<4, 2>(s, V, W, H, G, mp, wgs, slab);
    case 44: return launch_fused_t<4, 4>(s, V, W, H, G, mp, wgs, slab);
  }
  return PMF_EINVAL;
}
#endif  // PMF_FUSED_KERNEL_ONLY
