// colgemm_lab.hip -- stand-alone bench of the W^T V partial-slab kernels (NMFALS / tiled NMF update_h at cfg3:
// V 262 144 x 1 024, W 262 144 x 64) for quick iteration on the kernel structure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef WPAD
#define WPAD 4
#endif
#include "../pymf_amd/csrc/pmf_dev.h"
#include "../pymf_amd/csrc/pmf_tiled.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#ifndef LAB_VARIANT
template <int NT, int MODE>
__global__ __launch_bounds__(256) void k_col2(const float* __restrict__ V, int64_t ldv, int np, const float* __restrict__ W, int64_t ldw, int64_t mp,
                                              int rows_per_chunk, float* __restrict__ slab) {
  constexpr int KP = 16 * NT;
  constexpr int ST = (NT + 3) / 4;
  constexpr int WQ = NT >= 4 ? NT / 4 : 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int64_t r_begin = (int64_t)blockIdx.x * rows_per_chunk;
  int64_t r_end = r_begin + rows_per_chunk;
  if (r_end > mp) r_end = mp;
  const int c0 = blockIdx.y * 256 + 64 * wv;
  const bool pact = c0 < np;
  const bool sact = MODE != 3 && blockIdx.y == 0;      // MODE 3: MFMAs only, no S
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
  struct Operands { f32x4 w[WQ][4]; f32x4 v[4]; };
  const float* Vl = V + (int64_t)(4 * kq) * ldv + c0l + 4 * i;
  const float* Wl = W + (int64_t)(4 * kq) * ldw + NT * i;
  const int64_t r_last = r_end - 16;
  auto ld_v = [&](int64_t r, int j, Operands& o) { o.v[j] = *reinterpret_cast<const f32x4*>(Vl + (r + j) * ldv); };
  auto ld_w = [&](int64_t r, int j, Operands& o) {
#pragma unroll
    for (int q = 0; q < WQ; ++q) o.w[q][j] = *reinterpret_cast<const f32x4*>(Wl + (r + j) * ldw + 4 * q);
  };
  auto step = [&](const Operands& cur, Operands& nxt, int64_t rn) {
    rn = rn < r_end ? rn : r_last;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE != 2 && MODE != 3) ld_v(rn, j, nxt);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 1) { P[0][j] += cur.v[j]; P[1][j] += cur.w[0][j]; continue; }
#pragma unroll
      for (int mt = 0; mt < NT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) P[mt][nt] = mfma16(cur.w[mt / 4][j][mt % 4], cur.v[j][nt], P[mt][nt]);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE != 2 && MODE != 3) ld_w(rn, j, nxt);
      __builtin_amdgcn_sched_barrier(0);
      if (sact) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int st = 0; st < ST; ++st) {
            float b = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              if (nt == wv + 4 * st) b = cur.w[nt / 4][j][nt % 4];
            S[mt][st] = mfma16(cur.w[mt / 4][j][mt % 4], b, S[mt][st]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  Operands o0, o1, o2, o3;
#pragma unroll
  for (int j = 0; j < 4; ++j) { ld_v(r_begin, j, o0); ld_w(r_begin, j, o0); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { ld_v(r_begin + 16, j, o1); ld_w(r_begin + 16, j, o1); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { ld_v(r_begin + 32, j, o2); ld_w(r_begin + 32, j, o2); }
  for (int64_t r = r_begin; r < r_end; r += 64) {      // rows_per_chunk % 64 == 0
    step(o0, o3, r + 48);
    step(o1, o0, r + 64);
    step(o2, o1, r + 80);
    step(o3, o2, r + 96);
  }
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
        *reinterpret_cast<f32x4*>(rowp + c0 + 4 * i) = o;
      }
      if (sact) {
#pragma unroll
        for (int st = 0; st < ST; ++st) {
          const int nt = wv + 4 * st;
          if (nt < NT) rowp[np + NT * i + nt] = S[mt][st][jj];
        }
      }
    }
}
#endif

// Variant S: the mirror of k_rowgemm_stream -- V fragments straight into registers one 64-row stage ahead, the requests
// interleaved with the MFMAs; the W rows of a stage go through LDS once per workgroup (instead of four waves x four column
// panels fetching them from L2), one barrier per stage.
template <int NT, bool WITH_S>
__global__ __launch_bounds__(256, 2) void k_col3(const float* __restrict__ V, int64_t ldv, int np, const float* __restrict__ W, int64_t ldw, int64_t mp,
                                                 int rows_per_chunk, float* __restrict__ slab) {
  constexpr int KP = 16 * NT;
  constexpr int ST = (NT + 3) / 4;
  constexpr int SR = 64;                               // rows per stage
#ifndef WPAD
#define WPAD 4
#endif
  constexpr int WLD = KP + WPAD;                       // padded row of the W stage in LDS (floats)
  constexpr int WCH = SR * (KP / 4) / 256;             // 16-byte pieces of a W stage per thread
  static_assert(NT == 4, "lab: NT = 4");
  extern __shared__ __attribute__((aligned(16))) float sw[];
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
  f32x4 va0[4][4], va1[4][4];                          // [step of 16 rows][j]: V[r + 16 t + 4 kq + j][c0 + 4 i ..]
  f32x4 pw[WCH];
  auto load_w = [&](int s) {
    const int ss = s < nst ? s : nst - 1;
#pragma unroll
    for (int q = 0; q < WCH; ++q) {
      const int id = tid + 256 * q;
      pw[q] = *reinterpret_cast<const f32x4*>(W + (r_begin + (int64_t)ss * SR + id / (KP / 4)) * ldw + 4 * (id % (KP / 4)));
    }
  };
  auto store_w = [&](float* buf, int s) {
    const bool live = s < nst;                         // a stage beyond the chunk multiplies by zeros
#pragma unroll
    for (int q = 0; q < WCH; ++q) {
      const int id = tid + 256 * q;
      const f32x4 v = live ? pw[q] : f32x4{0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(buf + (id / (KP / 4)) * WLD + 4 * (id % (KP / 4))) = v;
    }
  };
  auto stage = [&](int s, f32x4 (&va)[4][4], f32x4 (&van)[4][4]) {
    const float* wb = sw + (s & 1) * (SR * WLD);
    const int sn = s + 1 < nst ? s + 1 : nst - 1;
    const float* Vn = Vl + (int64_t)sn * SR * ldv;
    __syncthreads();
    f32x4 a4 = *reinterpret_cast<const f32x4*>(wb + (4 * kq) * WLD + NT * i);   // bases NT i .. NT i + 3 of row 4 kq + j
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 a4n = a4;
        if (4 * t + j < 15) {                            // the next step's W fragment: its LDS round trip runs under these MFMAs
          const int tn = (4 * t + j + 1) >> 2, jn = (4 * t + j + 1) & 3;
          a4n = *reinterpret_cast<const f32x4*>(wb + (16 * tn + 4 * kq + jn) * WLD + NT * i);
        }
        van[t][j] = *reinterpret_cast<const f32x4*>(Vn + (int64_t)(16 * t + j) * ldv);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) P[mt][nt] = mfma16(a4[mt], va[t][j][nt], P[mt][nt]);
        if (sact) {
#pragma unroll
          for (int mt = 0; mt < NT; ++mt)
#pragma unroll
            for (int st = 0; st < ST; ++st) {
              float b = 0.f;
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                if (nt == wv + 4 * st) b = a4[nt];
              S[mt][st] = mfma16(a4[mt], b, S[mt][st]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        a4 = a4n;
      }
    store_w(sw + ((s + 1) & 1) * (SR * WLD), s + 1);
    load_w(s + 2);
  };
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) va0[t][j] = *reinterpret_cast<const f32x4*>(Vl + (int64_t)(16 * t + j) * ldv);
  load_w(0);
  store_w(sw, 0);
  load_w(1);
  for (int s = 0; s < nst; s += 2) {
    stage(s, va0, va1);
    stage(s + 1, va1, va0);
  }
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
        *reinterpret_cast<f32x4*>(rowp + c0 + 4 * i) = o;
      }
      if (sact) {
#pragma unroll
        for (int st = 0; st < ST; ++st) {
          const int nt = wv + 4 * st;
          if (nt < NT) rowp[np + NT * i + nt] = S[mt][st][jj];
        }
      }
    }
}

__global__ void fillk(float* p, size_t n, unsigned seed) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = u01_from(seed, i); }

int main(int argc, char** argv) {
  const int64_t m = argc > 1 ? atoll(argv[1]) : 262144;
  const int np = argc > 2 ? atoi(argv[2]) : 1024;
  constexpr int NT = 4, KP = 64;
  const int rpc = 1024, nchunks = (int)(m / rpc);
  float *V, *W, *S1, *S2;
  const size_t slab_elems = (size_t)nchunks * KP * (np + KP);
  CK(hipMalloc(&V, m * np * 4)); CK(hipMalloc(&W, m * KP * 4)); CK(hipMalloc(&S1, slab_elems * 4)); CK(hipMalloc(&S2, slab_elems * 4));
  fillk<<<(m * np + 255) / 256, 256>>>(V, m * np, 1); fillk<<<(m * KP + 255) / 256, 256>>>(W, m * KP, 2);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double flop = 2.0 * m * np * KP;
  dim3 grid(nchunks, (np + 255) / 256);
#define RUN(label, kern, out)                                                                       \
  for (int it = 0; it < 6; ++it) {                                                                  \
    CK(hipEventRecord(e0));                                                                         \
    hipLaunchKernelGGL((kern), grid, dim3(256), 0, 0, V, (int64_t)np, np, W, (int64_t)KP, m, rpc, out); \
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());                                             \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));                                                 \
    if (it >= 3) printf("%-22s %.3f ms  %.1f TFLOP/s (P only)\n", label, ms, flop / ms / 1e9);      \
  }
  RUN("k_colgemm<4>", k_colgemm<NT>, S1)
  RUN("k_col2<4> loads only", (k_col2<NT, 1>), S2)
  RUN("k_col2<4> MFMAs only", (k_col2<NT, 2>), S2)
  RUN("k_col2<4> MFMAs only, no S", (k_col2<NT, 3>), S2)
  RUN("k_col2<4>", (k_col2<NT, 0>), S2)
  {
    const size_t smem3 = (size_t)2 * 64 * (KP + WPAD) * sizeof(float);
    for (int it = 0; it < 6; ++it) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_col3<NT, true>), grid, dim3(256), smem3, 0, V, (int64_t)np, np, W, (int64_t)KP, m, rpc, S2);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 3) printf("%-22s %.3f ms  %.1f TFLOP/s (P only)\n", "k_col3<4,S>", ms, flop / ms / 1e9);
    }
  }
  std::vector<float> h1(slab_elems), h2(slab_elems);
  CK(hipMemcpy(h1.data(), S1, slab_elems * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h2.data(), S2, slab_elems * 4, hipMemcpyDeviceToHost));
  double md = 0; for (size_t q = 0; q < slab_elems; ++q) md = fmax(md, fabs(h1[q] - h2[q]) / fmax(1.0, fabs(h1[q])));
  printf("max rel diff of the slabs %.2e\n", md);
  return 0;
}
template __global__ void k_colgemm_stream<4, true>(const float*, int64_t, int, const float*, int64_t, int64_t, int, float*, int64_t, int);
template __global__ void k_colgemm_stream<4, false>(const float*, int64_t, int, const float*, int64_t, int64_t, int, float*, int64_t, int);
template __global__ void k_colgemm_stream<8, false>(const float*, int64_t, int, const float*, int64_t, int64_t, int, float*, int64_t, int);
