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

constexpr int PMF_IPC_MAX_RANKS = 8;
constexpr size_t PMF_IPC_MAX_BYTES = (size_t)256 << 10;     // payloads up to 256 KiB take this path
constexpr int PMF_IPC_MAX_WGS = 64;
constexpr unsigned long long PMF_IPC_WAIT_TICKS = 30ull * 100000000ull;
constexpr size_t pmf_ipc_flags_offset(int nranks) { return (size_t)2 * nranks * PMF_IPC_MAX_BYTES; }
constexpr size_t pmf_ipc_area_bytes(int nranks) {
  return pmf_ipc_flags_offset(nranks) + (size_t)2 * nranks * PMF_IPC_MAX_WGS * sizeof(unsigned);
}

struct IpcPeers {
  char* area[PMF_IPC_MAX_RANKS];     // every rank's receive area as mapped in THIS process (area[me] is the local one)
  int me, nranks;
};

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
  __threadfence_system();
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
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > wait_ticks) { ok = 0; break; }
    }
  }
  __syncthreads();
  if (!ok) { if (tid == 0) atomicExch(err, 1); return; }
  __threadfence_system();
  const char* base = pr.area[me] + (size_t)slot * N * PMF_IPC_MAX_BYTES;
  for (int64_t e = lo + tid; e < hi; e += 256) {
    T s = (T)0;       // (0 + p_0 + p_1 + ...: the order, and the sign of a zero, of the host transport's sum)
    for (int r = 0; r < N; ++r) s += __builtin_nontemporal_load(reinterpret_cast<const T*>(base + (size_t)r * PMF_IPC_MAX_BYTES) + e);
    p[e] = s;
  }
}
