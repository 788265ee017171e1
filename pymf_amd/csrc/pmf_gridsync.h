// pmf_gridsync.h -- what the workgroups of ONE launch need to hand data to each other (gfx950):
// a grid barrier on sharded arrival counters and the write-through (`sc1`) loads / stores of the
// handed-off bytes, plus the distributed sum of the per-workgroup slabs built on them.
//
// Why not __threadfence() + cooperative_groups::grid().sync(): the per-XCD L2s are not coherent with
// each other, so an agent-scope release writes back every dirty line of the XCD's L2 (the W rows the
// pass has just stored: several microseconds) and the software grid.sync() of ROCm 7.2 costs 26 us at
// 256 workgroups.  Here every handed-off byte is stored `sc1` (write-through, dropped from the L2) and
// loaded `sc1` (never served by the CU's L1), the storing waves drain their stores (s_waitcnt
// vmcnt(0)) before ONE lane of the workgroup adds to an arrival counter, and the consumer polls that
// counter with `sc1` loads and crosses a workgroup barrier before its first load of the bytes
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility", the row of
// agent-scope atomic adds polled by global_load_dword sc1).  No fence instruction is issued.
//
// The counters are monotonic inside a launch (barrier number b completes when the shards add up to
// b * gridDim.x) and are put back to zero by the workgroup that leaves the launch last.  Every wait is
// bounded (PMF_SYNC_TIMEOUT_TICKS of the 100 MHz realtime counter): a launch whose workgroups cannot
// all be resident -- another process holding CUs -- raises the abort word and ends instead of hanging;
// the first barrier of a launch is a census taken BEFORE anything is written, so such a launch has no
// side effects.
#pragma once
#include "pmf_dev.h"

typedef int pmf_i32x4 __attribute__((ext_vector_type(4)));

constexpr int PMF_SYNC_SHARDS = 8;            // arrival counters, one 128-byte line each
constexpr int PMF_SYNC_STRIDE = 32;           // unsigned words per line
constexpr int PMF_SYNC_EXIT = PMF_SYNC_SHARDS * PMF_SYNC_STRIDE;        // word index: workgroups that have left
constexpr int PMF_SYNC_ABORT = PMF_SYNC_EXIT + PMF_SYNC_STRIDE;         // word index: != 0 -> a wait timed out
constexpr int PMF_SYNC_WORDS = PMF_SYNC_ABORT + PMF_SYNC_STRIDE;
constexpr unsigned long long PMF_SYNC_TIMEOUT_TICKS = 300000000ull;     // 3 s of s_memrealtime (100 MHz)

// Raw buffer resource over [p, p + 2 GiB): the buffer forms of the loads / stores take the cache
// policy as an immediate (aux 16 = sc1) and are tracked by the compiler's waitcnt insertion.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pmf_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 pmf_ld16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const pmf_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void pmf_st16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pmf_i32x4, v), r, (int)byte_off, 0, 16);
}
__device__ __forceinline__ void pmf_st4_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long pmf_realtime() { return __builtin_amdgcn_s_memrealtime(); }

// Barrier number `bar` (1, 2, ...) of this launch.  Called by ALL threads of every workgroup, behind
// the stores the barrier publishes.  ok_lds: one LDS word of the caller.  Returns false when the launch
// has to end (time-out here or in another workgroup).
__device__ __forceinline__ bool pmf_grid_barrier(unsigned* w, unsigned bar, volatile int* ok_lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's stores have left the CU
  __syncthreads();                                        // ... and those of the workgroup's other waves
  const int tid = threadIdx.x;
  if (tid == 0)
    __hip_atomic_fetch_add(w + (blockIdx.x % PMF_SYNC_SHARDS) * PMF_SYNC_STRIDE, 1u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
  if (tid < 64) {
    const unsigned target = bar * gridDim.x;
    const unsigned long long t0 = pmf_realtime();
    int ok = 1;
    for (;;) {
      unsigned v = 0;
      if (tid < PMF_SYNC_SHARDS)
        v = __hip_atomic_load(w + tid * PMF_SYNC_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (tid == PMF_SYNC_SHARDS)
        v = __hip_atomic_load(w + PMF_SYNC_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned sum = 0;
#pragma unroll
      for (int s = 0; s < PMF_SYNC_SHARDS; ++s) sum += (unsigned)__builtin_amdgcn_readlane((int)v, s);
      const unsigned ab = (unsigned)__builtin_amdgcn_readlane((int)v, PMF_SYNC_SHARDS);
      if (sum >= target) break;
      if (ab != 0u || pmf_realtime() - t0 > PMF_SYNC_TIMEOUT_TICKS) {
        if (tid == 0) __hip_atomic_store(w + PMF_SYNC_ABORT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (tid == 0) *ok_lds = ok;
  }
  __syncthreads();                                        // between the poll and every load of the bytes
  return *ok_lds != 0;
}

// Last thing a workgroup does in a launch that used pmf_grid_barrier: the one that leaves last puts the
// counters back to zero for the next launch (stream order makes that visible).  The abort word stays.
__device__ __forceinline__ void pmf_grid_leave(unsigned* w) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(w + PMF_SYNC_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gridDim.x - 1) {
      for (int s = 0; s < PMF_SYNC_SHARDS; ++s)
        __hip_atomic_store(w + s * PMF_SYNC_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(w + PMF_SYNC_EXIT, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- distributed sum of the tile-major slabs ---------------------------------------------------
// A slab is `ntiles` tiles of 64 lanes x float4 (the accumulator layout of the one-pass kernels).  A
// chunk = 16 consecutive float4 of one tile; workgroup b sums chunks b, b + gridDim.x, ... over all
// slabs in float64: thread (g = tid / 16, e = tid % 16) adds slabs g, g + 16, ... of element e, the 16
// partial sums are combined in the order of g -- the order k_reduce_slabs_tiles uses, so both give the
// same bits.  The sums go to `pst` (one slab, written sc1) and, through `scatter`, wherever else the
// caller wants them (the row-major (P | S) image).  red: 16 * 16 * 4 doubles of LDS.  256 threads.
template <typename Scatter>
__device__ __forceinline__ void pmf_reduce_slabs_dist(const float* slab, int nslabs, int ntiles, float* pst,
                                                      double* red, Scatter scatter) {
  const int tid = threadIdx.x, g = tid >> 4, e = tid & 15;
  const __amdgpu_buffer_rsrc_t rs = pmf_rsrc(slab);
  const unsigned slab_bytes = (unsigned)ntiles * 1024u;
  for (int ch = blockIdx.x; ch < ntiles * 4; ch += gridDim.x) {
    const unsigned off = (unsigned)ch * 256u + (unsigned)e * 16u;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int c = g;
    for (; c + 48 < nslabs; c += 64) {                    // four loads in flight per thread
      const f32x4 a = pmf_ld16_sc1(rs, (unsigned)c * slab_bytes + off);
      const f32x4 b = pmf_ld16_sc1(rs, (unsigned)(c + 16) * slab_bytes + off);
      const f32x4 d = pmf_ld16_sc1(rs, (unsigned)(c + 32) * slab_bytes + off);
      const f32x4 f = pmf_ld16_sc1(rs, (unsigned)(c + 48) * slab_bytes + off);
      s0 += (double)a[0]; s1 += (double)a[1]; s2 += (double)a[2]; s3 += (double)a[3];
      s0 += (double)b[0]; s1 += (double)b[1]; s2 += (double)b[2]; s3 += (double)b[3];
      s0 += (double)d[0]; s1 += (double)d[1]; s2 += (double)d[2]; s3 += (double)d[3];
      s0 += (double)f[0]; s1 += (double)f[1]; s2 += (double)f[2]; s3 += (double)f[3];
    }
    for (; c < nslabs; c += 16) {
      const f32x4 a = pmf_ld16_sc1(rs, (unsigned)c * slab_bytes + off);
      s0 += (double)a[0]; s1 += (double)a[1]; s2 += (double)a[2]; s3 += (double)a[3];
    }
    double* mine = red + (g * 16 + e) * 4;
    mine[0] = s0; mine[1] = s1; mine[2] = s2; mine[3] = s3;
    __syncthreads();
    if (tid < 64) {
      const int comp = tid >> 4;                          // component `comp` of element e
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += red[(q * 16 + e) * 4 + comp];
      const float v = (float)t;
      const int tile = ch >> 2, lane = 16 * (ch & 3) + e;
      pmf_st4_sc1(pst + ((size_t)tile * 64 + lane) * 4 + comp, v);
      scatter(tile, lane, comp, v);
    }
    __syncthreads();                                      // red is reused by the next chunk
  }
}
