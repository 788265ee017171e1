// pmf_ipc.h -- one-shot all-reduce over IPC-mapped receive buffers ("one-shot P2P over the full mesh").
//
// The per-iteration exchange of the row-sharded factorize() loop is ONE sum of (W^T V | W^T W): k (n + k) floats, 80 KiB
// at 1 048 576 x 256, k = 64 (pymf/nmf.py:124-125 is what gets sharded).  At that size a ring or tree collective is pure
// latency (several dependent hops); on a fully connected xGMI node every rank can instead WRITE its partial straight
// into every peer's memory and every rank then adds the N partials itself:
//
//   * every rank owns a receive area [2 slots][N ranks][PMF_IPC_MAX_BYTES] plus flags [2][N][PMF_IPC_MAX_WGS], exported
//     with hipIpcGetMemHandle and mapped by every peer (hipIpcOpenMemHandle) -- a peer on another GPU reaches it over
//     its xGMI link, a peer process on the same GPU through the shared L2;
//   * ONE kernel per rank: workgroup g owns a slice of the payload; it stores its slice into slot [seq & 1][me] of
//     every peer (and of itself), makes the stores visible (system-scope fence), raises flag [seq & 1][me][g] = seq
//     at every peer, then waits until its own flags [seq & 1][r][g] show seq for all r and adds the N slices IN RANK
//     ORDER into the payload -- every rank forms the same sum in the same order, so the result (and with it H) is
//     bit-identical on all ranks, as with the host transport (which also adds in rank order);
//   * no barrier across workgroups or ranks beyond those flags: slice g of rank a only ever waits for slice g of the
//     peers.  Two slots suffice: a rank can be at most one exchange ahead of a peer (it cannot finish exchange s + 1
//     before that peer has raised its flags of s + 1, i.e. has finished reading s).
// A wait is bounded (wait_ticks of the 100 MHz counter; PMF_IPC_WAIT_TICKS = 30 s in the loops -- ranks may be seconds
// apart on the host side, e.g. one still reading its data --, 2 s in the self-test); a peer that never arrives raises
// *err instead of hanging the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pmf_dev.h"

constexpr int PMF_IPC_MAX_RANKS = 8;
constexpr size_t PMF_IPC_MAX_BYTES = (size_t)256 << 10;     // payloads up to 256 KiB take this path
constexpr int PMF_IPC_MAX_WGS = 128;   // flags per (slot, rank): one per workgroup of the pushing kernel (k_ipc_allreduce: <= 64; the
                                       // folded exchange: one per (P | S) tile of k_reduce_slabs_tiles, <= 74 on the one-pass shapes)
constexpr int PMF_IPC_AR_MAX_WGS = 64; // grid cap of k_ipc_allreduce
constexpr unsigned long long PMF_IPC_WAIT_TICKS = 30ull * 100000000ull;
constexpr size_t pmf_ipc_flags_offset(int nranks) { return (size_t)2 * nranks * PMF_IPC_MAX_BYTES; }
constexpr size_t pmf_ipc_area_bytes(int nranks) {
  return pmf_ipc_flags_offset(nranks) + (size_t)2 * nranks * PMF_IPC_MAX_WGS * sizeof(unsigned);
}

struct IpcPeers {
  char* area[PMF_IPC_MAX_RANKS];     // every rank's receive area as mapped in THIS process (area[me] is the local one)
  int me, nranks;
};

// ---- the two halves of the exchange as device helpers (round 5): a PRODUCER kernel pushes what it has just computed
// straight into every peer's receive area, a CONSUMER kernel waits for the peers' flags in its prologue and adds the N
// partials in rank order while it loads its operands -- the exchange is then no launch of its own (k_reduce_slabs_tiles /
// k_nmf_h_gram).  Same slots, same flags, same sequence counter, same order of additions as k_ipc_allreduce below.

// `v` to float index `idx` of this rank's slot [seq & 1][me] in every rank's receive area (the peers first, myself last)
__device__ __forceinline__ void ipc_push_f32(const IpcPeers& pr, unsigned seq, int64_t idx, float v) {
  const size_t slot_off = ((size_t)(seq & 1u) * pr.nranks + pr.me) * PMF_IPC_MAX_BYTES;
  for (int d = 1; d <= pr.nranks; ++d) {
    const int r = (pr.me + d) % pr.nranks;
    __builtin_nontemporal_store(v, reinterpret_cast<float*>(pr.area[r] + slot_off) + idx);
  }
}
// after ALL threads of the workgroup have pushed: every storing wave drains its stores (vmcnt counts a store until its write is
// acknowledged), the workgroup meets, and the lanes that raise flag g at the ranks do so with ONE system-scope release (the
// write-back in front of the flag store) -- not a fence in each of the workgroup's sixteen waves (MI355X_MICROARCH.md: a release
// costs 1.7-6.5 us and every fencing wave pays it)
__device__ __forceinline__ void ipc_raise(const IpcPeers& pr, unsigned seq, int g) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((int)threadIdx.x < pr.nranks) {
    unsigned* f = reinterpret_cast<unsigned*>(pr.area[threadIdx.x] + pmf_ipc_flags_offset(pr.nranks)) +
                  ((size_t)(seq & 1u) * pr.nranks + pr.me) * PMF_IPC_MAX_WGS + g;
    __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// consumer: every thread of the workgroup calls it; waits (bounded) until flags 0 .. nflags-1 of ALL ranks show seq.
// false: a peer did not arrive (the caller raises *err and returns).  `ok` is a __shared__ int of the caller.
__device__ __forceinline__ bool ipc_wait_all(const IpcPeers& pr, unsigned seq, int nflags, unsigned long long wait_ticks, int* ok) {
  if (threadIdx.x == 0) *ok = 1;
  __syncthreads();
  const unsigned* fl = reinterpret_cast<const unsigned*>(pr.area[pr.me] + pmf_ipc_flags_offset(pr.nranks)) +
                       (size_t)(seq & 1u) * pr.nranks * PMF_IPC_MAX_WGS;
  const unsigned long long t0 = wall_clock64();
  for (int q = threadIdx.x; q < nflags * pr.nranks; q += blockDim.x) {
    const unsigned* f = fl + (size_t)(q / nflags) * PMF_IPC_MAX_WGS + (q % nflags);
    // (relaxed system-scope polls: a cache-bypassing load each; the ONE acquire follows below)
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > wait_ticks) { *ok = 0; break; }
    }
  }
  __syncthreads();                                   // every flag has been seen
  if (!*ok) return false;
  if (threadIdx.x < 64) {                            // one wave acquires for the CU (its L1 is the CU's), the others wait for it
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  return true;
}
// the N partials of 4 consecutive floats at float index idx, added in rank order (0 + p_0 + p_1 + ...: the order, and the
// sign of a zero, of k_ipc_allreduce and of the host transport)
__device__ __forceinline__ f32x4 ipc_sum4(const IpcPeers& pr, unsigned seq, int64_t idx) {
  const char* base = pr.area[pr.me] + (size_t)(seq & 1u) * pr.nranks * PMF_IPC_MAX_BYTES;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < pr.nranks; ++r)
    s += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)r * PMF_IPC_MAX_BYTES + idx * sizeof(float)));
  return s;
}
__device__ __forceinline__ float ipc_sum1(const IpcPeers& pr, unsigned seq, int64_t idx) {
  const char* base = pr.area[pr.me] + (size_t)(seq & 1u) * pr.nranks * PMF_IPC_MAX_BYTES;
  float s = 0.f;
  for (int r = 0; r < pr.nranks; ++r)
    s += __builtin_nontemporal_load(reinterpret_cast<const float*>(base + (size_t)r * PMF_IPC_MAX_BYTES) + idx);
  return s;
}

// The split form on a plain buffer (pmf_ipc_selftest: the helpers above exactly as the slab-reduce / H-step kernels use them,
// a producer grid of many workgroups -- flags beyond k_ipc_allreduce's 64 -- and a consumer grid of a few 1024-thread ones).
__global__ __launch_bounds__(256) void k_ipc_fold_push(const float* __restrict__ p, int64_t count, IpcPeers pr, unsigned seq) {
  const int64_t per = (count + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < count ? lo + per : count;
  for (int64_t e = lo + threadIdx.x; e < hi; e += 256) ipc_push_f32(pr, seq, e, p[e]);
  ipc_raise(pr, seq, blockIdx.x);
}
__global__ __launch_bounds__(1024) void k_ipc_fold_pull(float* __restrict__ out, int64_t count, IpcPeers pr, unsigned seq, int nflags,
                                                        int* __restrict__ err, unsigned long long wait_ticks) {
  __shared__ int ok;
  if (!ipc_wait_all(pr, seq, nflags, wait_ticks, &ok)) { if (threadIdx.x == 0) atomicExch(err, 1); return; }
  const int64_t c4 = count / 4;
  for (int64_t q = (int64_t)blockIdx.x * 1024 + threadIdx.x; q < c4; q += (int64_t)gridDim.x * 1024)
    *reinterpret_cast<f32x4*>(out + 4 * q) = ipc_sum4(pr, seq, 4 * q);
  for (int64_t e = 4 * c4 + blockIdx.x * 1024 + threadIdx.x; e < count; e += (int64_t)gridDim.x * 1024) out[e] = ipc_sum1(pr, seq, e);
}

// T = float or double; p[count] is this rank's partial on entry and the all-rank sum on exit.
template <typename T>
__global__ __launch_bounds__(256) void k_ipc_allreduce(T* __restrict__ p, int64_t count, IpcPeers pr, unsigned seq, int* __restrict__ err,
                                                       unsigned long long wait_ticks) {
  const int g = blockIdx.x, tid = threadIdx.x;
  const int N = pr.nranks, me = pr.me, slot = (int)(seq & 1u);
  const int64_t per = ((count + gridDim.x - 1) / gridDim.x + 3) & ~(int64_t)3;
  const int64_t lo = (int64_t)g * per, hi = lo + per < count ? lo + per : count;
  const size_t slot_off = ((size_t)slot * N + me) * PMF_IPC_MAX_BYTES;
  // 1. my slice into every rank's receive area (the peers first, myself last)
  for (int d = 1; d <= N; ++d) {
    const int r = (me + d) % N;
    T* dst = reinterpret_cast<T*>(pr.area[r] + slot_off);
    for (int64_t e = lo + tid; e < hi; e += 256) __builtin_nontemporal_store(p[e], dst + e);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (every storing wave drains; the release rides on the flag stores below)
  __syncthreads();
  // 2. raise my flag of this slice at every rank
  if (tid < N) {
    unsigned* f = reinterpret_cast<unsigned*>(pr.area[tid] + pmf_ipc_flags_offset(N)) + ((size_t)slot * N + me) * PMF_IPC_MAX_WGS + g;
    __hip_atomic_store(f, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 3. wait for every rank's flag of this slice (bounded), then add the slices in rank order
  __shared__ int ok;
  if (tid == 0) ok = 1;
  __syncthreads();
  if (tid < N) {
    const unsigned* f = reinterpret_cast<const unsigned*>(pr.area[me] + pmf_ipc_flags_offset(N)) + ((size_t)slot * N + tid) * PMF_IPC_MAX_WGS + g;
    const unsigned long long t0 = wall_clock64();            // the 100 MHz constant-rate counter
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > wait_ticks) { ok = 0; break; }
    }
  }
  __syncthreads();
  if (!ok) { if (tid == 0) atomicExch(err, 1); return; }
  if (tid < 64) {                                    // one wave acquires for the CU, the others wait for it
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  const char* base = pr.area[me] + (size_t)slot * N * PMF_IPC_MAX_BYTES;
  for (int64_t e = lo + tid; e < hi; e += 256) {
    T s = (T)0;       // (0 + p_0 + p_1 + ...: the order, and the sign of a zero, of the host transport's sum)
    for (int r = 0; r < N; ++r) s += __builtin_nontemporal_load(reinterpret_cast<const T*>(base + (size_t)r * PMF_IPC_MAX_BYTES) + e);
    p[e] = s;
  }
}
