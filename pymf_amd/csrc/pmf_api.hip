// pmf_api.hip -- host side of libpymf_hip.so: the C ABI declared in include/pymf_hip.h.
//
// One pmf_ctx = one GPU, one HIP stream, optionally one RCCL communicator.
// Device layout (all float32, zero padded, row-major):
//   V  [mp][np]   mp = m rounded up to 64, np = n rounded up to 64
//   W  [mp][KP]   KP = 16*NT, NT in {1,2,4,8} (k <= 128)
//   H  [KP][np]
//   G  [KP][KP]   H H^T          PS [KP][np+KP]  (W^T V | W^T W)
// Zero padding is exactly neutral for all three update rules (a padded W column
// or H row/column stays 0 through every multiplicative step; SNMF/NMFALS put 1 on
// the padded diagonal of the k x k systems).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/pymf_hip.h"
#include "pmf_dev.h"
#include "pmf_ipc.h"
#include "pmf_small.h"
#include "pmf_tiled.h"
#include "pmf_fused_api.h"  // the one-pass kernels themselves: pmf_fused_tu.hip
#include "pmf_nnls_api.h"   // the sub-problem kernels themselves: pmf_nnls_tu.hip
#include "pmf_inv.h"
#include "pmf_csr.h"
#include "pmf_nndsvd.h"
#include "pmf_topk.h"

namespace {

constexpr int PMF_HGRAM_MAX_WGS = 64;

std::string g_create_error;

// Launch sites that can be bracketed by HIP events (pmf_profile_enable): ONE of them, the dominant
// m-sized kernel of the path the context takes, is recorded at a time (choose_stat_site).
enum { SITE_NONE = 0, SITE_FUSED, SITE_ROWGEMM_W, SITE_NNQP_W, SITE_MATERIALIZE, SITE_CSR_PASS };

struct KernelStat {
  std::string name = "none";
  int site = SITE_NONE;
  double flops = 0.0, bytes = 0.0, exec_flops = 0.0;
  std::vector<hipEvent_t> ev;   // pairs
  size_t used = 0;              // events recorded since reset
  int every = 1;                // pmf_set_option("profile_every", N): only every N-th launch of the site carries events -- a pair costs
                                // the loop ~5 us (the queue processes two more packets and the dispatch's completion signal): 8 % of a
                                // 60 us iteration when every launch is timed (tools/loop_probe.py, profiles/r05_experiments.md)
  int64_t seen = 0;             // launches of the site since reset
  bool open = false;            // stat_begin recorded, stat_end to follow
};

}  // namespace

struct pmf_ctx {
  int algo = 0;
  int64_t m = 0, n = 0;
  int k = 0, device = 0, rank = 0, nranks = 1;
  int nb = 1;                   // > 1: num_bases > 128 (NMF): KP = 128 nb, bases handled in blocks of 128
  float* dW2 = nullptr;         // ... Den = W (H H^T), [mp][KP] (dW1 holds Num = V H^T)
  float* dWideT = nullptr;      // chunk result of a product over more than PMF_WIDE_K columns
  int64_t wide_cap = 0;
  float *dWideN = nullptr, *dWideD = nullptr;   // ... Num and Den of the W rules there (wide_update_w_rows)
  int64_t wide_nd_cap = 0;
  int64_t mp = 0;
  int np = 0, KP = 0, NT = 0;
  hipStream_t stream = nullptr;
  ncclComm_t comm = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  float *dV = nullptr, *dW = nullptr, *dH = nullptr, *dG = nullptr, *dPS = nullptr;
  float *dSlab = nullptr, *dW1 = nullptr, *dGinvT = nullptr;
  float* dMT = nullptr;         // SNMF: M^T = inv(H H^T) H, [KP][np] (k_snmf_mt)
  double* dGinvD = nullptr;     // SNMF: inv(H H^T) in float64, [KP][KP]
  // Gram-space SNMF loop (snmf_gram_iteration): C = V^T V over all ranks' rows, and the float64 M^T, P
  double *dC = nullptr, *dMTd = nullptr, *dPd = nullptr;
  // SNMF (num_bases <= 128): H in float64 on the device (pmf_inv.h, round 6) -- dH is its float32 rounding.  hd_synced: k_hd_sync
  // has compared the two in THIS API call (need() clears it); ps_f64: (P | S) of the Gram-space iteration at hand are in dPd / dSd
  double *dHd = nullptr, *dSd = nullptr, *dHdSnap = nullptr;
  bool hd_synced = false, hd_force = false, ps_f64 = false;   // hd_force: H was replaced through a float32 entry point
  bool psd_fresh = false;      // dPd / dSd are the float64 (P | S) of the CURRENT W (set by a Gram-space iteration, for the error behind it; an API entry clears it)
  int opt_snmf_h64 = 1;
  double* dCslabs = nullptr;    // k_csr_gram: per-workgroup images of C's upper triangle (two 64-bit fixed-point limbs per entry)
  unsigned* dVmaxBits = nullptr; // ... and the bit pattern of the largest |v| (the limbs' grids)
  bool c_valid = false;         // dC holds the all-rank V^T V of the current V
  int opt_snmf_gram = -1;       // pmf_set_option("snmf_gram"): -1 auto, 0 never, 1 whenever possible, 2 = 1 + W written in every iteration
  bool w_implicit = false;      // the loop ran in Gram space: dW is stale, W = V M with the M at hand (materialize_w)
  // snmf_gram = 2 on CSR data: W = V M of iteration i is written on a stream of its own BESIDE the k x n sized kernels of
  // iteration i + 1 (they never read W); M is double buffered (dW1, dW1 + np KP) and the write launch leaves a few
  // workgroup slots free so that the small kernels can be placed while it runs (w_pipe_* below, materialize_w)
  hipStream_t w_stream = nullptr;
  hipEvent_t ev_mt[2] = {nullptr, nullptr}, ev_w[2] = {nullptr, nullptr};
  bool ev_w_pending[2] = {false, false};
  int64_t w_pipe_it = 0;        // writes enqueued so far: buffer parity
  int opt_w_pipe = 32;          // pmf_set_option("snmf_w_pipe"): workgroup slots the write launch leaves free; 0 = in stream order
  float* dD = nullptr;          // RNMF: D = S - V (rnmf.py:102,111), [mp][np]
  bool s_valid = false;         // RNMF: D has been formed (update_s ran)
  double rnmf_err2 = -1.0;      // RNMF: sum((V - W H)^2) from the last update_s (all ranks)
  double *dGd = nullptr, *dPart = nullptr, *dScal = nullptr;
  double* dGramPart = nullptr;  // k_gram_splitk: per-slice partial Gram matrices, [8][KP][KP]
  unsigned* dGramTickets = nullptr;   // k_gram_splitk: one per tile, [tiles] the tile count of the fused Gram + inverse, [tiles + 1] k_reduce_slabs_inv's
  // NMFALS at 64 bases, round 6 experiment: the k x k chain of a half step as ONE launch -- the workgroup that completes the
  // Hessian inverts it (pmf_inv.h: k_gram_splitk<float, true>, k_reduce_slabs_inv); pmf_set_option("fuse_chain", 1 | 2) turns
  // the W / H half step's fused form on
  int opt_fuse_chain = 0;    // (measured: NOT faster -- profiles/r06_experiments.md; kept as an A/B knob with its bit-equality test)
  bool want_inv = false, chain_prepared = false;
  float* dGpart = nullptr;      // k_nmf_h_gram: per-workgroup partial G, [PMF_HGRAM_MAX_WGS][KP][KP]
  float* dHsnap = nullptr;      // pmf_snapshot_h: H, then G, then the partial Gs
  bool hsnap_valid = false, hsnap_g_valid = false;
  int hsnap_g_parts = 0;
  double* dT1part = nullptr;    // ... and partial <P, H_new>
  unsigned* dTicket = nullptr;  // ... arrival counter (the kernel resets it)
  // free-running pmf_factorize loop: device-side error history and stop flag
  double* dFerr = nullptr; int64_t ferr_cap = 0;
  int* dStop = nullptr;         // [0] 0 run / 1 converged / 2 identity cancels, [1] iteration
  int* dWarm = nullptr;         // k_nnqp: warm start allowed (k_spd_unique)
  const int* stop_arg = nullptr;   // what the loop kernels get: dStop while free-running, else NULL
  // free-running loop: the error / convergence test of iteration conv_iter (>= 0) is still to be evaluated, from
  // conv_ntt pairs of trace terms at conv_tt; the next one-pass launch does it in its prologue (FusedCtl)
  int conv_iter = -1, conv_ntt = 0;
  const double* conv_tt = nullptr;
  double conv_eps = 0.0;
  // CSR V (SNMF sparse path)
  int64_t* dIndptr = nullptr; int32_t* dIndices = nullptr; float* dVals = nullptr; int64_t nnz = 0;
  bool v_csr = false;
  bool csr_dense = false;       // CSR data with num_bases > 128: a dense image in dV serves the data paths (no CSR kernel at that width)
  int* dSing = nullptr;         // SNMF: raised by the inverse kernels when H H^T has a zero pivot (check_singular)
  double* dQp = nullptr;        // k_nnqp_big (NMFALS, num_bases > 64): per-workgroup inverse images
  double* dBinv = nullptr;      // k_nnqp_quad: B = inv(HA), [KP][KP] float64
  int* dDefer = nullptr;        // k_nnqp_quad<16>: problems left to the 32-slot frame
  int64_t defer_cap = 0;
  int* dNbig = nullptr;         // [2 sites][3 + 2]: rotating counters of k_nnqp_quad (QuadCtl: nbig x 3, dcount x 2)
  int64_t quad_calls[2] = {0, 0};
  unsigned long long* dQstat = nullptr;   // k_nnqp_quad, W half steps: [2 frames][4] running totals (pmf_nnqp_counters)
  double* dY0 = nullptr;        // k_nnqp_wave: inv(HA) f of every problem of a half step
  int64_t y0_cap = 0;
  int opt_nnqp_wave = 1;        // pmf_set_option("nnqp_wave"): 64 < num_bases <= 128 on the wave-per-problem block-pivoting kernel
  int opt_nnqp_frame16 = 1;     // pmf_set_option("nnqp_frame16"): the 16-slot frame first (three waves per SIMD)
  int opt_nnqp_count = 0;       // pmf_set_option("nnqp_count"): the counting instantiations of k_nnqp_quad (pmf_nnqp_counters)
  void* dStage = nullptr;       // staging area of the host <-> device transport (upload_rows / download_rows)
  size_t stage_cap = 0;
  float* dWsnap = nullptr;      // pmf_snapshot_w: the W before a step that may fail
  bool wsnap_valid = false;
  int opt_nndsvd_topk = -1;     // pmf_set_option("nndsvd_topk"): -1 by size, 1 the filtered subspace iteration, 0 full Jacobi
  int nndsvd_products = 0;      // products with the Gram matrix the last top-k solve took
  int opt_colgemm_stream = 1;   // pmf_set_option("colgemm_stream"): W^T V partials on k_colgemm_stream where it applies
  int opt_resid_resident = 1;   // residual pass with H resident in LDS (k_resid_res) where it fits
  int resid_parts = 0;          // float64 partials the last residual pass left in dPart
  int64_t dpart_cap = 0;        // doubles dPart holds
  int opt_rowgemm_stream = 1;   // pmf_set_option("rowgemm_stream"): plain products with a long contraction on k_rowgemm_stream
  int opt_nnqp_quad = 1;        // pmf_set_option("nnqp_quad"): num_bases <= 64 on the sixteen-lanes-per-problem kernel
  double *dInvA = nullptr, *dInvB = nullptr;   // k_inverse_spd_big: the two images of the elimination, [KP][KP]
  int nchunks = 0, rows_per_chunk = 0;
  int fused_wgs = 0;            // >0: fused one-pass kernel available for this shape
  int fused_wgs_hidden = 0;     // pmf_set_option("force_tiled", 1) parks fused_wgs / fused8 here: every path then takes the
  bool fused8_hidden = false;   // any-shape two-pass kernels (k_rowgemm / k_colgemm) -- test and measurement aid
  std::string path_hidden;
  bool fused8 = false;          // ... and it is the cooperative form (pmf_coop.h: 64 < k <= 128, or k <= 64 with n > 256)
  int coop_bt = 0, coop_rb = 0; // its base tiles per wave / row blocks per tile
  bool have_v = false, have_w = false, have_h = false, g_valid = false;
  int g_parts = 0;              // > 0 (with g_valid): G = sum of that many partials in dGpart, dG is stale
  int trace_parts = 0;          // > 0 (with trace_ready): the trace terms are that many pairs in dT1part
  bool gram_partial_ok = false; // pmf_factorize: the next consumer of G is the fused kernel
  bool want_hess = false, gd_is_s = false;   // NMFALS H half step: the reduce writes dGd = W^T W itself (reduce_slabs)
  bool ps_valid = false;        // dPS = (W^T V | W^T W) of the CURRENT W, summed over all ranks
  bool num_valid = false;       // dW1 holds Num = V H^T of the current V, H (fixed-H loops, NMF)
  bool fixed_h_loop = false;    // pmf_factorize running compute_w without compute_h for > 1 iteration
  bool want_trace = false;      // pmf_factorize with PMF_COMPUTE_ERR: let the H-step kernel emit the trace terms
  bool trace_ready = false;     // dScal[2..3] already hold <P,H>, <S,HH^T> for the current W, H
  bool vnorm_valid = false;
  bool vnorm_local_valid = false;   // dScal[6] = sum(V^2) over this rank's rows (formed behind the upload)
  double vnorm2 = 0.0;          // ||V||_F^2 over all ranks
  double lamb_w = 0.0, lamb_h = 0.0;   // BNMF penalty weights (bnmf.py:84-85,118-119)
  // streamed V (pmf_stream_*): row tiles pass through two device buffers, V is never resident
  float* dTile[2] = {nullptr, nullptr};
  int64_t tile_cap = 0;                      // rows per tile buffer (multiple of 64)
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
  double* dPSacc = nullptr;                  // float64 (P | S) accumulated over the tiles of a pass
  double* dStAcc = nullptr;                  // [0] sum v^2, [1] sum (v - (W H))^2 over the tiles
  bool st_active = false, st_vnorm_pending = false;
  uint32_t st_flags = 0;
  int64_t st_rows_seen = 0;
  int st_tiles = 0;
  // one-shot all-reduce over IPC-mapped receive areas (pmf_ipc.h): payloads <= PMF_IPC_MAX_BYTES
  IpcPeers ipc{};                            // ipc.nranks > 1: ready
  bool ipc_exported = false;
  unsigned long long ipc_wait_ticks = PMF_IPC_WAIT_TICKS;
  int ipc_nranks_ready = 0;                  // ranks mapped by pmf_ipc_import (ipc.nranks = 0 while the path is switched off)
  int ipc_export_nranks = 0;                 // the rank count pmf_ipc_export sized the receive area for
  float *dIpcTestA = nullptr, *dIpcTestB = nullptr;   // pmf_ipc_selftest's payloads, allocated by pmf_ipc_export (no allocation -- nothing
                                                      // that can fail locally -- between the self-test's collectives)
  unsigned ipc_seq = 0;
  std::atomic<int> abort_flag{0};            // pmf_abort: another host thread asks the running pmf_factorize loop to return early
  // the folded exchange (round 5): inside pmf_factorize's one-pass loop the push rides on k_reduce_slabs_tiles and the wait +
  // rank-ordered sum on k_nmf_h_gram's prologue -- no launch for the exchange (pmf_set_option("fold_exchange", 0): the
  // k_ipc_allreduce launch of round 4 instead)
  int opt_fold = 1;
  bool fold_loop = false;                    // set by nmf_fused_iteration around its two launches
  unsigned fold_seq = 0;                     // != 0: k_reduce_slabs_tiles has pushed exchange fold_seq, the next k_nmf_h_gram consumes it
  int fold_flags = 0;                        // tiles (= flags) of that push
  unsigned long long* dIpcWait = nullptr;    // [2]: ticks of the 100 MHz counter the consumer spent waiting, exchanges counted
  int64_t fold_calls = 0;
  int* dIpcErr = nullptr;
  int64_t coll_seen = 0;
  int64_t ipc_calls = 0, rccl_calls = 0, host_calls = 0;   // which transport the cross-rank sums took (pmf_collective_name)
  pmf_host_allreduce_fn host_ar = nullptr;   // host transport for the cross-rank sums (pmf_set_host_allreduce)
  void* host_ar_user = nullptr;
  std::vector<unsigned char> ar_buf;
  bool profile = false;
  std::vector<hipEvent_t> coll_ev;           // event pairs around the per-iteration collective (allreduce_ps)
  size_t coll_used = 0;
  bool host_ar_only() const { return host_ar != nullptr && ipc.nranks <= 1 && comm == nullptr; }   // (blocking host round trips: nothing to time on the stream)
  double last_loop_ms = 0.0;
  KernelStat stat;
  std::string err;
  std::string path;
};

namespace {

int fail(pmf_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_error = msg;
  return code;
}

#define HIPCHK(c, expr)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail((c), e_ == hipErrorOutOfMemory ? PMF_ENOMEM : PMF_EHIP,                 \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                     \
  } while (0)

#define NCCLCHK(c, expr)                                                                  \
  do {                                                                                    \
    ncclResult_t r_ = (expr);                                                             \
    if (r_ != ncclSuccess)                                                                \
      return fail((c), PMF_ENCCL, std::string(#expr) + ": " + ncclGetErrorString(r_));    \
  } while (0)

#define PMFCHK(expr)                 \
  do {                               \
    int rc_ = (expr);                \
    if (rc_ != PMF_OK) return rc_;   \
  } while (0)

int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

int ensure_dv(pmf_ctx* c);

int csr_ps(pmf_ctx* c);
int csr_w(pmf_ctx* c, hipStream_t stream, const float* Mbuf, int reserve);
int materialize_w(pmf_ctx* c);

template <typename T>
int dalloc(pmf_ctx* c, T** p, size_t count) {
  HIPCHK(c, hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)));
  HIPCHK(c, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), c->stream));
  return PMF_OK;
}

// CSR kernels serve the data paths (SNMF, num_bases <= 128); wider contexts keep a dense image of the CSR rows
static inline bool use_csr(const pmf_ctx* c) { return c->v_csr && !c->csr_dense; }

int ensure_dv(pmf_ctx* c) {
  if (c->dV) return PMF_OK;
  return dalloc(c, &c->dV, (size_t)c->mp * c->np);
}

// ---- profiling of the dominant kernel -------------------------------------------------
void stat_begin(pmf_ctx* c, int site) {
  if (!c->profile || c->stat.site != site) return;
  KernelStat& s = c->stat;
  s.open = false;
  if (s.seen++ % s.every != 0) return;
  if (s.used + 2 > s.ev.size()) {
    for (int q = 0; q < 2; ++q) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      s.ev.push_back(e);
    }
  }
  (void)hipEventRecord(s.ev[s.used], c->stream);   // profiling aid: a failed record only loses a sample
  s.open = true;
}
// The next pair of events of site `site`, to be attached to a dispatch (hipExtLaunchKernelGGL); nullptr when not profiling.
void stat_pair(pmf_ctx* c, int site, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  if (!c->profile || c->stat.site != site) return;
  KernelStat& s = c->stat;
  if (s.seen++ % s.every != 0) return;
  if (s.used + 2 > s.ev.size()) {
    for (int q = 0; q < 2; ++q) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      s.ev.push_back(e);
    }
  }
  *e0 = s.ev[s.used]; *e1 = s.ev[s.used + 1];
  s.used += 2;
}
void stat_end(pmf_ctx* c, int site) {
  if (!c->profile || c->stat.site != site) return;
  KernelStat& s = c->stat;
  if (!s.open || s.used + 2 > s.ev.size()) return;
  s.open = false;
  (void)hipEventRecord(s.ev[s.used + 1], c->stream);
  s.used += 2;
}

// ---- kernel launch helpers --------------------------------------------------------------
// Grid of the element-wise kernels (256 threads, grid-stride loops): one thread per element up to 2^30 threads -- a launch of
// more than 2^32 threads wraps without an error (found with a 36 Mi x 256 matrix, tests/sweeps/huge_probe.py).
static inline unsigned elem_grid(int64_t count) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>((count + 255) / 256, (int64_t)1 << 22));
}

template <int NT, int EPI>
int launch_rowgemm(pmf_ctx* c, const float* A, int64_t lda, int kdimA, const float* B, int64_t ldb,
                   float* W, const float* G, float* C, int64_t rows_p = -1, int64_t mvalid = -1, int64_t ldc = 0) {
  if (rows_p < 0) { rows_p = c->mp; mvalid = c->m; }
  if (ldc == 0) ldc = 16 * NT;
  const float lamb = (float)c->lamb_w;
  if constexpr (EPI == EPI_STORE || EPI == EPI_NMF_W || EPI == EPI_BNMF_W || EPI == EPI_RNMF_W) {
    if (c->opt_rowgemm_stream && kdimA % 128 == 0) {
      // long contraction: A straight into registers, requests interleaved with the MFMAs (pmf_tiled.h)
      constexpr int RB = NT <= 4 ? 4 : 2;
      const int ntiles = (int)(rows_p / (16 * RB));
      // persistent workgroups, two per CU of a 256-CU part (a fixed count; at 128 bases with the Den product one group each)
      const int ngroups = (ntiles + 3) / 4;
      const bool single = (EPI != EPI_STORE) && NT > 4;
      const unsigned grid = (unsigned)(single ? ngroups : std::min(ngroups, 512));
      const size_t ssm = rowgemm_stream_smem_bytes<NT, EPI, false>();
      if (ssm > 64 * 1024) {
        static bool sattr_dev[PMF_MAX_DEVICES] = {};
        bool& sattr = sattr_dev[pmf_current_device()];
        if (!sattr) {
          HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowgemm_stream<NT, RB, EPI>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)ssm));
          sattr = true;
        }
      }
      hipLaunchKernelGGL((k_rowgemm_stream<NT, RB, EPI>), dim3(grid), dim3(256), ssm, c->stream, A, lda, kdimA, B, ldb, W, G, C, ldc, lamb, mvalid,
                         c->k, ntiles, (int64_t)(16 * NT));
      HIPCHK(c, hipGetLastError());
      return PMF_OK;
    }
  }
  const size_t smem = rowgemm_smem_bytes<NT>();
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rowgemm<NT, EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_done = true;
  }
  const int ntiles = (int)(rows_p / 64);
  const int tpw = ntiles >= 8192 ? 8 : ntiles >= 2048 ? 4 : ntiles >= 1024 ? 2 : 1;   // consecutive tiles per workgroup
  hipLaunchKernelGGL((k_rowgemm<NT, EPI>), dim3((unsigned)((ntiles + tpw - 1) / tpw)), dim3(256), smem, c->stream,
                     A, lda, kdimA, B, ldb, W, G, C, ldc, lamb, mvalid, c->k, ntiles, tpw);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

template <int EPI>
int rowgemm_one(pmf_ctx* c, const float* A, int64_t lda, int kdimA, const float* B, int64_t ldb,
                float* W, const float* G, float* C, int64_t rows_p = -1, int64_t mvalid = -1) {
  if (c->nb > 1) {            // num_bases > 128: the plain product in blocks of 128 bases, C is [.][KP]
    if (EPI != EPI_STORE) return fail(c, PMF_EINVAL, "rowgemm: only the plain product runs in base blocks");
    for (int b = 0; b < c->nb; ++b)
      PMFCHK((launch_rowgemm<8, EPI_STORE>(c, A, lda, kdimA, B + (size_t)b * 128 * ldb, ldb, nullptr, nullptr, C + b * 128,
                                           rows_p, mvalid, c->KP)));
    return PMF_OK;
  }
  switch (c->NT) {
    case 1: return launch_rowgemm<1, EPI>(c, A, lda, kdimA, B, ldb, W, G, C, rows_p, mvalid);
    case 2: return launch_rowgemm<2, EPI>(c, A, lda, kdimA, B, ldb, W, G, C, rows_p, mvalid);
    case 4: return launch_rowgemm<4, EPI>(c, A, lda, kdimA, B, ldb, W, G, C, rows_p, mvalid);
    case 8: return launch_rowgemm<8, EPI>(c, A, lda, kdimA, B, ldb, W, G, C, rows_p, mvalid);
  }
  return fail(c, PMF_EINVAL, "bad NT");
}

// A product over the columns of V is ONE accumulation chain of kdim / 4 MFMA steps per output element, and the fp32 MFMA
// does not round its running sum to nearest: the chain loses a fraction of about 1.5e-16 * steps^2 of the sum (measured on
// uniform data, tests/sweeps/wide_scan.py: V H^T biased by -2e-7 at 32 768 columns, -3.2e-6 at 131 072, -4.2e-5 at 524 288,
// -1.5e-4 at 10^6 -- W comes out scaled by that factor and H by its inverse, the fit itself is unaffected).  Products over
// more than PMF_WIDE_K columns are therefore formed in chunks of PMF_WIDE_K columns whose results are added in float32
// (round to nearest): the bias stays at the 65 536-column level (1e-6) whatever n.  Shapes up to 65 536 columns run as before.
constexpr int PMF_WIDE_K = 65536;

template <int EPI>
int rowgemm(pmf_ctx* c, const float* A, int64_t lda, int kdimA, const float* B, int64_t ldb,
            float* W, const float* G, float* C, int64_t rows_p = -1, int64_t mvalid = -1) {
  if constexpr (EPI == EPI_STORE) {
    if (kdimA > PMF_WIDE_K) {
      const int64_t rp = rows_p < 0 ? c->mp : rows_p;
      const int64_t count = rp * c->KP;
      if (c->wide_cap < count) {
        if (c->dWideT) { (void)hipFree(c->dWideT); c->dWideT = nullptr; c->wide_cap = 0; }
        PMFCHK(dalloc(c, &c->dWideT, (size_t)count));
        c->wide_cap = count;
      }
      for (int k0 = 0; k0 < kdimA; k0 += PMF_WIDE_K) {
        const int kc = std::min(PMF_WIDE_K, kdimA - k0);
        PMFCHK(rowgemm_one<EPI_STORE>(c, A + k0, lda, kc, B + k0, ldb, W, G, k0 == 0 ? C : c->dWideT, rows_p, mvalid));
        if (k0 > 0) {
          hipLaunchKernelGGL(k_acc_f32, dim3(elem_grid(count / 4)), dim3(256), 0, c->stream, C, c->dWideT, count);
          HIPCHK(c, hipGetLastError());
        }
      }
      return PMF_OK;
    }
  }
  return rowgemm_one<EPI>(c, A, lda, kdimA, B, ldb, W, G, C, rows_p, mvalid);
}

// The W rules of NMF / BNMF / RNMF over more than PMF_WIDE_K columns: Num = X H^T in chunks (above), Den = W G by a small
// kernel, the rule element by element -- what the one-launch forms (update rule as the product's epilogue) cannot do in chunks.
int wide_update_w_rows(pmf_ctx* c, const float* X, float* Wr, int64_t rows_p, int64_t mvalid) {
  const int64_t count = rows_p * c->KP;
  if (c->wide_nd_cap < count) {
    if (c->dWideN) { (void)hipFree(c->dWideN); c->dWideN = nullptr; }
    if (c->dWideD) { (void)hipFree(c->dWideD); c->dWideD = nullptr; }
    c->wide_nd_cap = 0;
    PMFCHK(dalloc(c, &c->dWideN, (size_t)count));
    PMFCHK(dalloc(c, &c->dWideD, (size_t)count));
    c->wide_nd_cap = count;
  }
  float *Num = c->dWideN, *Den = c->dWideD;
  PMFCHK(rowgemm<EPI_STORE>(c, X, c->np, c->np, c->dH, c->np, nullptr, nullptr, Num, rows_p, mvalid));
  hipLaunchKernelGGL(k_den_small, dim3(elem_grid(count)), dim3(256), 0, c->stream, Wr, c->dG, Den, rows_p, c->KP);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(k_nmf_w_elem, dim3(elem_grid(count)), dim3(256), 0, c->stream, Wr, Num, Den, count,
                     c->algo == PMF_ALGO_BNMF ? 1 : c->algo == PMF_ALGO_RNMF ? 2 : 0, (float)c->lamb_w, c->KP, mvalid, c->k);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

// Partials of (W^T X | W^T W) over row chunks into c->dSlab: X [rows_p][xn] (ldx), W [rows_p][.] (ldw), chunks of rpc rows.
// k_colgemm_stream where it applies (NT = 4, or NT = 8 without S; rpc a multiple of its stage), else k_colgemm.
template <int NT, bool WITH_S>
int launch_colgemm(pmf_ctx* c, const float* X, int64_t ldx, int xn, const float* W, int64_t ldw, int64_t rows_p, int rpc, int nch,
                   float* slab = nullptr) {
  if (!slab) slab = c->dSlab;
  const dim3 grid((unsigned)nch, X ? (unsigned)((xn + 255) / 256) : 1u);
  constexpr int SR = NT == 4 ? 64 : 32;
  const bool stream_ok = c->opt_colgemm_stream && X != nullptr && rpc % SR == 0 && rows_p % SR == 0;
  const size_t smem = (size_t)2 * SR * (16 * NT + 4) * sizeof(float);
  if constexpr (NT == 4 || (NT == 8 && !WITH_S)) {
    if (stream_ok) {
      hipLaunchKernelGGL((k_colgemm_stream<NT, WITH_S>), grid, dim3(256), smem, c->stream, X, ldx, xn, W, ldw, rows_p, rpc, slab,
                         (int64_t)xn + 16 * NT, 0);
      HIPCHK(c, hipGetLastError());
      return PMF_OK;
    }
  }
  if constexpr (NT == 8 && WITH_S) {
    // 64 < num_bases <= 128: with the S tiles k_colgemm<8> holds 192 accumulator registers; the stream kernel forms P and,
    // as a second product with W in V's place, S -- into the same slabs (columns [xn, xn + 128))
    if (stream_ok) {
      hipLaunchKernelGGL((k_colgemm_stream<8, false>), grid, dim3(256), smem, c->stream, X, ldx, xn, W, ldw, rows_p, rpc, slab,
                         (int64_t)xn + 128, 0);
      hipLaunchKernelGGL((k_colgemm_stream<8, false>), dim3((unsigned)nch, 1u), dim3(256), smem, c->stream, W, ldw, 128, W, ldw, rows_p, rpc,
                         slab, (int64_t)xn + 128, xn);
      HIPCHK(c, hipGetLastError());
      return PMF_OK;
    }
  }
  hipLaunchKernelGGL((k_colgemm<NT, WITH_S>), grid, dim3(256), 0, c->stream, X, ldx, xn, W, ldw, rows_p, rpc, slab);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

int colgemm_rows(pmf_ctx* c, const float* X, const float* W, int64_t rows_p, int rpc, int nch) {
  switch (c->NT) {
    case 1: return launch_colgemm<1, true>(c, X, c->np, c->np, W, c->KP, rows_p, rpc, nch);
    case 2: return launch_colgemm<2, true>(c, X, c->np, c->np, W, c->KP, rows_p, rpc, nch);
    case 4: return launch_colgemm<4, true>(c, X, c->np, c->np, W, c->KP, rows_p, rpc, nch);
    case 8: return launch_colgemm<8, true>(c, X, c->np, c->np, W, c->KP, rows_p, rpc, nch);
  }
  return fail(c, PMF_EINVAL, "bad NT");
}

int colgemm(pmf_ctx* c, bool with_v = true) {
  const float* Vp = with_v ? (c->algo == PMF_ALGO_RNMF ? c->dD : c->dV) : nullptr;
  return colgemm_rows(c, Vp, c->dW, c->mp, c->rows_per_chunk, c->nchunks);
}

int64_t ps_elems(const pmf_ctx* c) { return (int64_t)c->KP * (c->np + c->KP); }

bool multi_rank(const pmf_ctx* c);

int reduce_slabs(pmf_ctx* c, int nslabs) {
  const int64_t E = ps_elems(c);      // multiple of 4 (KP and np are multiples of 16)
  // NMFALS on one rank: the column QPs' Hessian S = W^T W leaves the same launch in float64 (with more ranks it has to come
  // from the ALL-REDUCED sums: k_hessian_from_ps behind the collective)
  const bool hess = c->want_hess && !multi_rank(c);
  c->chain_prepared = false;
  if (hess && c->want_inv && c->KP == 64 && (c->opt_fuse_chain & 2)) {
    // ... and the workgroup that finishes last inverts it: flag, patched Hessian and B = inv(HA) leave the same launch
    if (!c->dWarm) PMFCHK(dalloc(c, &c->dWarm, 1));
    if (!c->dBinv) PMFCHK(dalloc(c, &c->dBinv, (size_t)2 * c->KP * c->KP));
    if (!c->dGramTickets) PMFCHK(dalloc(c, &c->dGramTickets, (size_t)(c->KP / 16) * (c->KP / 16) + 2));
    hipLaunchKernelGGL(k_reduce_slabs_inv, dim3((unsigned)((E / 4 + 63) / 64)), dim3(256), 0, c->stream, c->dSlab, nslabs, E, c->dPS, c->dGd,
                       c->np, c->KP, c->k, c->dGramTickets + (c->KP / 16) * (c->KP / 16) + 1, c->dBinv, c->dWarm, c->dBinv + (size_t)c->KP * c->KP);
    HIPCHK(c, hipGetLastError());
    c->gd_is_s = true;
    c->chain_prepared = true;
    return PMF_OK;
  }
  hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((E / 4 + 63) / 64)), dim3(1024), 0, c->stream,
                     c->dSlab, nslabs, E, c->dPS, hess ? c->dGd : (double*)nullptr, c->np, c->KP, c->k);
  HIPCHK(c, hipGetLastError());
  if (hess) c->gd_is_s = true;
  return PMF_OK;
}

// Sum `count` floats (or doubles) at device pointer `p` over all ranks, in place, in stream order.
// Transport: the context's RCCL communicator (one ncclAllReduce on the library's stream); or, when the
// caller installed a host transport (pmf_set_host_allreduce: plumbing checks where the ranks cannot form
// an RCCL communicator, e.g. several ranks sharing one GPU), a blocking round trip through the host.
int allreduce_sum(pmf_ctx* c, void* p, size_t count, bool f64) {
  const size_t nbytes = count * (f64 ? sizeof(double) : sizeof(float));
  if (c->ipc.nranks > 1 && nbytes <= PMF_IPC_MAX_BYTES && count > 0) {
    // one kernel: every rank writes its partial into every peer's receive area and adds the N partials in rank order
    const unsigned seq = ++c->ipc_seq;
    const int64_t vec = (int64_t)(count + 1023) / 1024;                         // ~1024 elements per workgroup
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(vec, PMF_IPC_AR_MAX_WGS));
    if (f64) hipLaunchKernelGGL((k_ipc_allreduce<double>), dim3(grid), dim3(256), 0, c->stream, (double*)p, (int64_t)count, c->ipc, seq, c->dIpcErr, c->ipc_wait_ticks);
    else hipLaunchKernelGGL((k_ipc_allreduce<float>), dim3(grid), dim3(256), 0, c->stream, (float*)p, (int64_t)count, c->ipc, seq, c->dIpcErr, c->ipc_wait_ticks);
    HIPCHK(c, hipGetLastError());
    ++c->ipc_calls;
    return PMF_OK;
  }
  if (c->host_ar) {
    ++c->host_calls;
    const size_t bytes = count * (f64 ? sizeof(double) : sizeof(float));
    c->ar_buf.resize(bytes);
    HIPCHK(c, hipMemcpyAsync(c->ar_buf.data(), p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->host_ar(c->host_ar_user, c->ar_buf.data(), (int64_t)count, f64 ? 1 : 0) != 0)
      return fail(c, PMF_ENCCL, "the host all-reduce callback reported a failure");
    HIPCHK(c, hipMemcpyAsync(p, c->ar_buf.data(), bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PMF_OK;
  }
  if (c->comm) {
    ++c->rccl_calls;
    NCCLCHK(c, ncclAllReduce(p, p, count, f64 ? ncclDouble : ncclFloat, ncclSum, c->comm, c->stream));
  }
  return PMF_OK;
}

bool multi_rank(const pmf_ctx* c) { return c->comm != nullptr || c->host_ar != nullptr || c->ipc.nranks > 1; }

// a peer that never raised its flags (k_ipc_allreduce gave up after ipc_wait_ticks of polling: 30 s in the loops, 2 s in the self-test)
int ipc_check(pmf_ctx* c) {
  if (c->ipc.nranks <= 1 || !c->dIpcErr) return PMF_OK;
  int e = 0;
  HIPCHK(c, hipMemcpyAsync(&e, c->dIpcErr, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (e) {
    HIPCHK(c, hipMemsetAsync(c->dIpcErr, 0, sizeof(int), c->stream));
    return fail(c, PMF_ENCCL, "one-shot all-reduce: a peer rank did not arrive (its flags were not raised within the polling limit)");
  }
  return PMF_OK;
}

// The per-iteration collective: (W^T V | W^T W) summed over the ranks.  With pmf_profile_enable its launches are bracketed by
// HIP events of their own (pmf_collective_ms: what the exchange costs an iteration at N > 1, next to the dominant kernel).
int allreduce_ps(pmf_ctx* c) {
  // (timed only where the sum is a device operation on the stream: the one-shot kernel or ncclAllReduce -- a payload that
  //  falls back to the blocking host round trip has nothing for HIP events to bracket)
  const bool on_stream = (c->ipc.nranks > 1 && (size_t)ps_elems(c) * sizeof(float) <= PMF_IPC_MAX_BYTES) || (!c->host_ar && c->comm);
  const bool timed = c->profile && multi_rank(c) && on_stream && (c->coll_seen++ % c->stat.every == 0);   // sampled like the kernel's
  if (timed) {
    if (c->coll_used + 2 > c->coll_ev.size())
      for (int q = 0; q < 2; ++q) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) c->coll_ev.push_back(e); }
    if (c->coll_used + 2 <= c->coll_ev.size()) (void)hipEventRecord(c->coll_ev[c->coll_used], c->stream);
  }
  const int rc = allreduce_sum(c, c->dPS, (size_t)ps_elems(c), false);
  if (timed && c->coll_used + 2 <= c->coll_ev.size()) {
    (void)hipEventRecord(c->coll_ev[c->coll_used + 1], c->stream);
    c->coll_used += 2;
  }
  return rc;
}

// What a one-pass launch needs to know about the free-running loop around it; hands over (and clears) the
// pending convergence test.
FusedCtl take_fused_ctl(pmf_ctx* c) {
  FusedCtl ctl{};
  ctl.stop = c->stop_arg ? c->dStop : nullptr;
  ctl.conv_iter = -1;
  if (ctl.stop && c->conv_iter >= 0) {
    ctl.tt = c->conv_tt; ctl.ntt = c->conv_ntt; ctl.ferr = c->dFerr;
    ctl.vnorm2 = c->vnorm2; ctl.eps = c->conv_eps; ctl.nsamp = (double)c->n;
    ctl.conv_iter = c->conv_iter;
    c->conv_iter = -1;
  }
  return ctl;
}

// ---- CSR (SNMF) ----------------------------------------------------------------------------
template <int NT>
int launch_csr_w_blocks(pmf_ctx* c, hipStream_t stream, const float* Mbuf, int reserve) {
  const size_t mbytes = (size_t)c->np * c->KP * sizeof(float);
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csr_w_blocks<NT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_done = true;
  }
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  const int64_t nblk = c->mp / 16;
  const int64_t nwg_full = (nblk + 15) / 16;          // one workgroup (16 waves) per 16 blocks = 256 rows = one contiguous piece of W
  // Round 4 (tools/csrw_lab.hip): a NON-persistent grid -- every workgroup writes ONE contiguous 256-row piece of W and
  // leaves, the pieces swept through memory in dispatch order -- with M read from L2 (64 KiB, resident; no LDS image to
  // stage per workgroup) stores at 6.75 TB/s where 512 persistent workgroups striding through W reach 5.4 (a pure store
  // stream of that strided shape: 5.1-5.6; hipMemsetAsync: 6.45).  Taken when the grid is several waves of workgroups deep.
  if (nwg_full >= (int64_t)8 * cus && mbytes <= (size_t)1 << 20) {
    hipLaunchKernelGGL((k_csr_w_blocks<NT>), dim3((unsigned)nwg_full), dim3(1024), 0, stream, c->dIndptr, c->dIndices, c->dVals,
                       nblk, c->np, Mbuf, c->dW, 0);
    HIPCHK(c, hipGetLastError());
    return PMF_OK;
  }
  const int in_lds = mbytes <= 128 * 1024;
  const size_t smem = in_lds ? mbytes : 0;
  const int per_cu = smem <= 80 * 1024 ? 2 : 1;     // workgroups of 16 waves per CU
  const unsigned wgs = (unsigned)std::max<int64_t>(1, std::min<int64_t>(nwg_full, (int64_t)cus * per_cu - reserve));
  hipLaunchKernelGGL((k_csr_w_blocks<NT>), dim3(wgs), dim3(1024), smem, stream, c->dIndptr, c->dIndices, c->dVals,
                     nblk, c->np, Mbuf, c->dW, in_lds);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

// W = V M, M = H^T inv(H H^T) (np x KP) in dW1 (snmf_inverse formed it) -- or, for the pipelined write, in Mbuf on `stream`
int csr_w(pmf_ctx* c, hipStream_t stream = nullptr, const float* Mbuf = nullptr, int reserve = 0) {
  if (!stream) stream = c->stream;
  if (!Mbuf) Mbuf = c->dW1;
  switch (c->NT) {
    case 1: return launch_csr_w_blocks<1>(c, stream, Mbuf, reserve);
    case 2: return launch_csr_w_blocks<2>(c, stream, Mbuf, reserve);
    case 4: return launch_csr_w_blocks<4>(c, stream, Mbuf, reserve);
    case 8: return launch_csr_w_blocks<8>(c, stream, Mbuf, reserve);
  }
  return fail(c, PMF_EINVAL, "bad NT");
}

// The pipelined W write of the snmf_gram = 2 loop on CSR data.
bool w_pipe_on(const pmf_ctx* c) { return c->opt_snmf_gram == 2 && use_csr(c) && c->opt_w_pipe > 0 && (size_t)2 * c->np * c->KP <= (size_t)std::max<int64_t>(c->mp, c->np) * c->KP; }
float* w_pipe_mbuf(pmf_ctx* c, int64_t it) { return c->dW1 + (size_t)(it & 1) * c->np * c->KP; }
int w_pipe_init(pmf_ctx* c) {
  if (c->w_stream) return PMF_OK;
  HIPCHK(c, hipStreamCreateWithFlags(&c->w_stream, hipStreamNonBlocking));
  for (int b = 0; b < 2; ++b) {
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_mt[b], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_w[b], hipEventDisableTiming));
  }
  return PMF_OK;
}
// every write enqueued on the side stream has finished before anything later on the main stream runs
int w_pipe_join(pmf_ctx* c) {
  for (int b = 0; b < 2; ++b)
    if (c->ev_w_pending[b]) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_w[b], 0)); c->ev_w_pending[b] = false; }
  return PMF_OK;
}

int csr_ps(pmf_ctx* c) {   // slabs: S part by the dense W^T W kernel, P part by the CSR scatter
  const size_t smem = (size_t)c->np * c->KP * sizeof(float);
  if (smem > 160 * 1024) return fail(c, PMF_EINVAL, "CSR path: n * num_bases too large for the LDS accumulator");
  PMFCHK(colgemm(c, /*with_v=*/false));
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csr_p<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csr_p<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  if (c->KP <= 64)
    hipLaunchKernelGGL((k_csr_p<1>), dim3((unsigned)c->nchunks), dim3(256), smem, c->stream, c->dIndptr,
                       c->dIndices, c->dVals, c->mp, c->rows_per_chunk, c->KP, c->np, c->dW, c->dSlab);
  else
    hipLaunchKernelGGL((k_csr_p<2>), dim3((unsigned)c->nchunks), dim3(256), smem, c->stream, c->dIndptr,
                       c->dIndices, c->dVals, c->mp, c->rows_per_chunk, c->KP, c->np, c->dW, c->dSlab);
  HIPCHK(c, hipGetLastError());
  return reduce_slabs(c, c->nchunks);
}

// SNMF keeps H in float64 on the device (num_bases <= 128; pmf_set_option("snmf_h64", 0): the float32 H of rounds 1-5)
static inline bool snmf_h64(const pmf_ctx* c) { return c->algo == PMF_ALGO_SNMF && c->nb == 1 && c->opt_snmf_h64 != 0; }

// dHd exists and agrees with dH: entries whose rounding is not the float32 H any more are replaced by the widened float32 value
int ensure_hd(pmf_ctx* c) {
  if (!c->dHd) {
    PMFCHK(dalloc(c, &c->dHd, (size_t)c->KP * c->np));
    PMFCHK(dalloc(c, &c->dSd, (size_t)c->KP * c->KP));
    c->hd_synced = false;
  }
  if (c->hd_synced) return PMF_OK;
  const int64_t E = (int64_t)c->KP * c->np;
  hipLaunchKernelGGL(k_hd_sync, dim3((unsigned)std::min<int64_t>((E + 255) / 256, 1024)), dim3(256), 0, c->stream, c->dH, c->dHd, E, c->hd_force ? 1 : 0);
  HIPCHK(c, hipGetLastError());
  c->hd_synced = true; c->hd_force = false;
  return PMF_OK;
}

int ensure_gram(pmf_ctx* c, double pad_diag) {
  if (c->g_valid && c->g_parts > 0) {            // k_nmf_h_gram left partial sums: add them up
    const int E = c->KP * c->KP;
    hipLaunchKernelGGL(k_sum_gparts, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, c->stream, c->dGpart, c->g_parts, E, c->dG);
    HIPCHK(c, hipGetLastError());
    c->g_parts = 0;
  }
  if (c->g_valid) return PMF_OK;
  c->g_parts = 0;   // (a count left behind by an H step whose H has been replaced since: the partials in dGpart are that H's)
  const bool h64 = snmf_h64(c);
  if (h64) PMFCHK(ensure_hd(c));                 // SNMF: G = Hd Hd^T, the float64 H
  dim3 grid((unsigned)(c->KP / 16), (unsigned)(c->KP / 16));
  const int ks = c->np >= 2048 && c->np % 512 == 0 ? 8 : c->np >= 512 && c->np % 256 == 0 ? 4 : 1;   // column slices (wide H)
  if (ks > 1 && c->nb == 1) {
    if (!c->dGramPart) {
      PMFCHK(dalloc(c, &c->dGramPart, (size_t)8 * c->KP * c->KP));
      PMFCHK(dalloc(c, &c->dGramTickets, (size_t)(c->KP / 16) * (c->KP / 16) + 2));
    }
    grid.z = (unsigned)ks;
    if (h64) hipLaunchKernelGGL(k_gram_splitk<double>, grid, dim3(256), 0, c->stream, c->dHd, (int64_t)c->np, c->np, c->KP, c->k, pad_diag, c->dG, c->dGd,
                                c->dGramPart, c->dGramTickets);
    else hipLaunchKernelGGL(k_gram_splitk<float>, grid, dim3(256), 0, c->stream, c->dH, (int64_t)c->np, c->np, c->KP, c->k, pad_diag, c->dG, c->dGd,
                            c->dGramPart, c->dGramTickets);
  } else {
    if (h64) hipLaunchKernelGGL(k_gram<double>, grid, dim3(256), 0, c->stream, c->dHd, (int64_t)c->np, c->np, c->KP, c->k,
                                pad_diag, c->dG, c->dGd);
    else hipLaunchKernelGGL(k_gram<float>, grid, dim3(256), 0, c->stream, c->dH, (int64_t)c->np, c->np, c->KP, c->k,
                            pad_diag, c->dG, c->dGd);
  }
  HIPCHK(c, hipGetLastError());
  c->g_valid = true;
  return PMF_OK;
}

int need(pmf_ctx* c, bool v, bool w, bool h) {
  if (!c) return PMF_EINVAL;
  c->hd_synced = false;        // (a new API call: whoever wrote the float32 H since the last one is noticed by k_hd_sync)
  c->psd_fresh = false;
  if (v && !c->have_v) return fail(c, PMF_EINVAL, "V has not been set (pmf_set_v_*)");
  if (w && !c->have_w) return fail(c, PMF_EINVAL, "W has not been set (pmf_set_w_f32)");
  if (h && !c->have_h) return fail(c, PMF_EINVAL, "H has not been set (pmf_set_h_f32)");
  HIPCHK(c, hipSetDevice(c->device));
  return PMF_OK;
}

// ---- NNDSVD initialisation (pymf/nndsvd.py:79-108; kernels and the closed form: pmf_nndsvd.h) ----
struct DevTemps {                 // scratch of one pmf_nndsvd_init call
  std::vector<void*> p;
  ~DevTemps() { for (void* q : p) (void)hipFree(q); }
};

template <typename T>
int talloc(pmf_ctx* c, DevTemps& t, T** out, size_t count) {
  PMFCHK(dalloc(c, out, count));
  t.p.push_back(*out);
  return PMF_OK;
}

// Ad [np][np] (float64) = V^T V of the dense V, summed over all ranks: 128 (or 64) Gram rows per pass of
// k_colgemm with the column block of V as its "W" operand (fp32 MFMA products, float64 slab sums).
// slab: gchunks * 128 * (np + 128) floats of scratch; rpc rows per chunk.
int gram_vtv(pmf_ctx* c, double* Ad, float* slab, int gchunks, int rpc) {
  const int np = c->np;
  // block row c0 against the columns from c0 on only (the matrix is symmetric: half the products), mirrored at the end
  for (int c0 = 0; c0 < np;) {
    const int wdt = (np - c0 >= 128) ? 128 : 64;
    const int xn = np - c0;
    if (wdt == 128) PMFCHK((launch_colgemm<8, false>(c, c->dV + c0, np, xn, c->dV + c0, np, c->mp, rpc, gchunks, slab)));
    else PMFCHK((launch_colgemm<4, false>(c, c->dV + c0, np, xn, c->dV + c0, np, c->mp, rpc, gchunks, slab)));
    // (k_gram_reduce: one thread per element walking the slabs one load at a time -- 0.2 ms per pass at 512 slabs)
    hipLaunchKernelGGL((k_reduce_slabs_block<double>), dim3((unsigned)(((int64_t)wdt * xn / 4 + 63) / 64)), dim3(1024), 0, c->stream, slab,
                       gchunks, wdt, xn + wdt, xn, Ad + (size_t)c0 * np + c0, (int64_t)np, 0);
    HIPCHK(c, hipGetLastError());
    c0 += wdt;
  }
  hipLaunchKernelGGL(k_mirror_upper_f64, dim3((unsigned)(((int64_t)np * np + 255) / 256)), dim3(256), 0, c->stream, Ad, np);
  HIPCHK(c, hipGetLastError());
  return allreduce_sum(c, Ad, (size_t)np * np, true);
}

// ---- top-k eigenpairs of the Gram matrix (pmf_topk.h) ----------------------------------------------
// C[M x N] = A[M x K] B, B stored [K][N] (or [N][K]: transb); M, N, K multiples of 16; float64 MFMA.
int dgemm64(pmf_ctx* c, const double* A, int64_t lda, const double* B, int64_t ldb, int K, double* C, int64_t ldc, int M, int N,
            bool transb) {
  const dim3 grid((unsigned)(N / 16), (unsigned)(M / 16));
  if (transb) hipLaunchKernelGGL((k_dgemm_mfma<true>), grid, dim3(64), 0, c->stream, A, lda, B, ldb, K, C, ldc, (float*)nullptr, (int64_t)0, (const int*)nullptr);
  else hipLaunchKernelGGL((k_dgemm_mfma<false>), grid, dim3(64), 0, c->stream, A, lda, B, ldb, K, C, ldc, (float*)nullptr, (int64_t)0, (const int*)nullptr);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

// eigen-decomposition of the nj x nj (nj even) symmetric A (leading dimension ld, as A2 and QT) on the device: evals
// (unsorted), rows of QT
int jacobi_eigh_dev(pmf_ctx* c, double* A, double* A2, double* QT, int ld, int nj, double* evals, int* sweeps_done) {
  const int64_t items = (int64_t)(nj / 2) * (nj / 2) + (int64_t)(nj / 2) * nj;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  const int64_t max_wgs = nj > 1024 ? cus : 64;      // one 1024-thread workgroup per CU at most (cooperative launch)
  const unsigned wgs = (unsigned)std::max<int64_t>(1, std::min<int64_t>(max_wgs, items / 4096));
  int ld_ = ld, nj_ = nj, sweeps_ = 40;
  void* args[] = {&A, &A2, &QT, &ld_, &nj_, &sweeps_, &evals, &sweeps_done};
  HIPCHK(c, hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_jacobi_eigh), dim3(wgs), dim3(1024), args,
                                       (unsigned)jacobi_smem_bytes(nj), c->stream));
  return PMF_OK;
}

// The k largest eigenpairs of G [np][np] (symmetric positive semi-definite, rows / columns >= n zero): rows 0 .. nl-1 of L
// ([round_up(k, 16)][np], zeroed by the caller) and ev[0 .. nl-1], descending as locked; nl <= k.  pmf_topk.h has the method.
int eigh_topk(pmf_ctx* c, DevTemps& tmp, double* G, int n, int np, int k, double* L, double* ev_dev, int* nl_out, int* products_out) {
  const int ld = np;
  const int kp16 = (int)round_up(k, 16);
  const int pblk = std::max(16, std::min(64, k / 2));
  const int s = (int)std::min<int64_t>(round_up(k + pblk, 16), (n / 16) * 16);
  if (s < 16 || s < k) return fail(c, PMF_EINVAL, "eigh_topk: the block does not fit the matrix");
  const int64_t cnt = (int64_t)s * ld;
  double *Ya, *Yb, *Yc, *Z, *D, *GQ, *T1, *S, *S2, *QTs, *Ug, *C1, *dth, *dev_ev, *dsc, *dres;
  int *dperm, *dinfo;
  for (double** q : {&Ya, &Yb, &Yc, &Z, &D, &GQ, &T1}) PMFCHK(talloc(c, tmp, q, (size_t)cnt));
  for (double** q : {&S, &S2, &QTs, &Ug}) PMFCHK(talloc(c, tmp, q, (size_t)s * s));
  PMFCHK(talloc(c, tmp, &C1, (size_t)s * kp16));
  PMFCHK(talloc(c, tmp, &dth, (size_t)kp16));
  PMFCHK(talloc(c, tmp, &dev_ev, (size_t)s));
  PMFCHK(talloc(c, tmp, &dsc, (size_t)s));
  PMFCHK(talloc(c, tmp, &dres, (size_t)s));
  PMFCHK(talloc(c, tmp, &dperm, (size_t)s));
  PMFCHK(talloc(c, tmp, &dinfo, 2));
  std::vector<double> th(s), hev(s), hsc(s), hres(s), thl;
  std::vector<int> perm(s);
  int nl = 0, products = 0;
  uint64_t seed = 0x9e3779b97f4a7c15ull;
  auto blocks = [](int64_t count) { return dim3((unsigned)((count + 255) / 256)); };
  auto fill_random = [&](double* Y, int r0, int r1) -> int {
    hipLaunchKernelGGL(k_topk_fill_random, blocks((int64_t)(r1 - r0) * ld), dim3(256), 0, c->stream, Y, r0, r1, ld, n, seed++);
    HIPCHK(c, hipGetLastError());
    return PMF_OK;
  };
  // D = (Y L^T) diag(theta or 1) L : the locked directions' part of Y (scaled: of A Y)
  auto locked_part = [&](const double* Y, bool scaled) -> int {
    PMFCHK(dgemm64(c, Y, ld, L, ld, np, C1, kp16, s, kp16, true));
    if (scaled) {
      hipLaunchKernelGGL(k_topk_scale_cols, blocks((int64_t)s * kp16), dim3(256), 0, c->stream, C1, s, kp16, kp16, dth);
      HIPCHK(c, hipGetLastError());
    }
    return dgemm64(c, C1, kp16, L, ld, kp16, D, ld, s, np, false);
  };
  // Zout = Y A'  (A' = A with the locked pairs deflated); returns with D = the deflation term when with_d
  auto apply = [&](const double* Y, double* Zout) -> int {
    ++products;
    PMFCHK(dgemm64(c, Y, ld, G, ld, np, Zout, ld, s, np, false));
    if (nl > 0) PMFCHK(locked_part(Y, true));
    return PMF_OK;
  };
  auto eigh_small = [&](double* A) -> int {                     // A (s x s) -> QTs rows, hev (host, unsorted)
    PMFCHK(jacobi_eigh_dev(c, A, S2, QTs, s, s, dev_ev, dinfo));
    HIPCHK(c, hipMemcpyAsync(hev.data(), dev_ev, (size_t)s * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PMF_OK;
  };
  auto ortho = [&](double*& Y) -> int {
    for (int attempt = 0; attempt < 4; ++attempt) {
      if (nl > 0)
        for (int rep = 0; rep < 2; ++rep) {
          PMFCHK(locked_part(Y, false));
          hipLaunchKernelGGL(k_topk_sub, blocks(cnt), dim3(256), 0, c->stream, Y, D, cnt);
          HIPCHK(c, hipGetLastError());
        }
      bool deficient = false;
      for (int rep = 0; rep < 2 && !deficient; ++rep) {
        PMFCHK(dgemm64(c, Y, ld, Y, ld, np, S, s, s, s, true));
        PMFCHK(eigh_small(S));
        double lmax = 0.0;
        for (int j = 0; j < s; ++j) lmax = std::max(lmax, hev[j]);
        if (!(lmax > 0.0) || !std::isfinite(lmax)) return fail(c, PMF_EHIP, "eigh_topk: the block collapsed");
        for (int j = 0; j < s; ++j) {
          if (!(hev[j] > 1e-24 * lmax)) { deficient = true; hsc[j] = 0.0; }
          else hsc[j] = 1.0 / std::sqrt(hev[j]);
        }
        HIPCHK(c, hipMemcpyAsync(dsc, hsc.data(), (size_t)s * sizeof(double), hipMemcpyHostToDevice, c->stream));
        PMFCHK(dgemm64(c, QTs, s, Y, ld, s, T1, ld, s, np, false));
        hipLaunchKernelGGL(k_topk_scale_rows, blocks(cnt), dim3(256), 0, c->stream, T1, s, ld, dsc);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));               // hsc is reused
        std::swap(Y, T1);
        if (deficient)                                            // dependent rows came out as zeros: new random ones, again
          for (int j = 0; j < s; ++j)
            if (hsc[j] == 0.0) PMFCHK(fill_random(Y, j, j + 1));
      }
      if (!deficient) return PMF_OK;
    }
    return fail(c, PMF_EHIP, "eigh_topk: could not orthonormalise the block");
  };
  // Rayleigh-Ritz on the orthonormal rows Y: GQ = Y A', T = Y GQ^T, rows rotated to the Ritz vectors, th descending
  auto rayleigh_ritz = [&](double*& Y) -> int {
    PMFCHK(apply(Y, GQ));
    if (nl > 0) {
      hipLaunchKernelGGL(k_topk_sub, blocks(cnt), dim3(256), 0, c->stream, GQ, D, cnt);
      HIPCHK(c, hipGetLastError());
    }
    PMFCHK(dgemm64(c, Y, ld, GQ, ld, np, S, s, s, s, true));
    PMFCHK(eigh_small(S));
    for (int j = 0; j < s; ++j) perm[j] = j;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return hev[a] > hev[b]; });
    for (int j = 0; j < s; ++j) th[j] = hev[perm[j]];
    HIPCHK(c, hipMemcpyAsync(dperm, perm.data(), (size_t)s * sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dev_ev, th.data(), (size_t)s * sizeof(double), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_topk_gather_rows, blocks((int64_t)s * s), dim3(256), 0, c->stream, QTs, (int64_t)s, dperm, Ug, (int64_t)s, s, s);
    HIPCHK(c, hipGetLastError());
    PMFCHK(dgemm64(c, Ug, s, Y, ld, s, T1, ld, s, np, false));
    std::swap(Y, T1);
    PMFCHK(dgemm64(c, Ug, s, GQ, ld, s, T1, ld, s, np, false));
    std::swap(GQ, T1);
    hipLaunchKernelGGL(k_topk_resid, dim3((unsigned)s), dim3(256), 0, c->stream, GQ, Y, ld, np, dev_ev, dres);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hres.data(), dres, (size_t)s * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PMF_OK;
  };

  hipLaunchKernelGGL(k_topk_symmetrise, blocks((int64_t)n * n), dim3(256), 0, c->stream, G, ld, n);
  HIPCHK(c, hipGetLastError());
  PMFCHK(fill_random(Ya, 0, s));
  PMFCHK(ortho(Ya));
  PMFCHK(rayleigh_ritz(Ya));
  const double scale = std::max(th[0], 1e-300);
  // A pair is locked when its residual is below 1e-11 of ITS OWN eigenvalue: the error of the vector is residual / gap, and
  // a tolerance relative to lambda_1 cannot be met by the dominant pair itself (its rounding floor is ~1e-12 lambda_1 at
  // n = 4608) while being too loose for the pairs of the bulk (1e-13 lambda_1 = 6e-7 against gaps of 0.05 there).
  double tol = 1e-11, prev_lead = 1e300;
  int stagnant = 0;
  constexpr int kMaxIter = 300, kMaxDeg = 40;
  for (int it = 0; it < kMaxIter && nl < k; ++it) {
    // ---- lock the leading converged pairs, in order ----
    const int need = k - nl;
    if (std::getenv("PMF_TOPK_DEBUG")) fprintf(stderr, "topk it %d: locked %d products %d th[0] %.6e th[need-1] %.6e th[s-1] %.6e res[0]/th %.2e res[need-1]/th %.2e\n", it, nl, products, th[0], th[std::min(need, s) - 1], th[s - 1], hres[0] / std::max(th[0], 1e-300), hres[std::min(need, s) - 1] / std::max(th[std::min(need, s) - 1], 1e-300));
    int nlock = 0;
    while (nlock < std::min(need, s) && (hres[nlock] <= tol * th[nlock] || th[nlock] <= 1e-14 * scale)) ++nlock;
    // the leading pair sits on its rounding floor (its residual no longer halves from one filter to the next; at
    // n = 16 384 the floor of the bulk pairs is 3e-10 of their eigenvalue: the deflated lambda_1 leaves eps lambda_1 behind):
    // take the floor as the tolerance
    const double lead = hres[0] / std::max(th[0], 1e-300);
    if (nlock == 0) {
      stagnant = (lead > 0.5 * prev_lead) ? stagnant + 1 : 0;
      if (stagnant >= 2 || it > kMaxIter - 3) {
        if (lead > 1e-8) return fail(c, PMF_EHIP, "pmf_nndsvd_init: the top-k eigen-solver stalled (residual " + std::to_string(lead) + " of the eigenvalue)");
        tol = std::max(tol, 2.0 * lead);
        stagnant = 0;
        prev_lead = 1e300;
        continue;
      }
    }
    prev_lead = nlock > 0 ? 1e300 : lead;
    if (nlock > 0) stagnant = 0;
    if (nlock > 0) {
      HIPCHK(c, hipMemcpyAsync(L + (size_t)nl * ld, Ya, (size_t)nlock * ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      for (int j = 0; j < nlock; ++j) thl.push_back(th[j]);
      nl += nlock;
      HIPCHK(c, hipMemcpyAsync(dth, thl.data(), (size_t)nl * sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (nl >= k) break;
      // the rest of the block moves up, fresh random rows behind it
      HIPCHK(c, hipMemcpyAsync(T1, Ya + (size_t)nlock * ld, (size_t)(s - nlock) * ld * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      std::swap(Ya, T1);
      PMFCHK(fill_random(Ya, s - nlock, s));
      PMFCHK(ortho(Ya));
      PMFCHK(rayleigh_ritz(Ya));
      continue;
    }
    // ---- Chebyshev filter: damp [0, cut], degree bounded by the dynamic range inside the block ----
    const double cut = std::max(th[s - 1], 1e-10 * scale), top = std::max(th[0], cut * (1.0 + 1e-12));
    const double e = 0.5 * cut, cc = 0.5 * cut;
    const double x_top = (top - cc) / e, x_k = (std::max(th[std::min(need, s) - 1], cut) - cc) / e;
    int deg = kMaxDeg;
    {
      const double g_top = std::acosh(std::max(x_top, 1.0)), g_k = std::acosh(std::max(x_k, 1.0));
      if (g_top - g_k > 0.0) deg = (int)std::max(2.0, std::min((double)kMaxDeg, std::floor(std::log(1e9) / (g_top - g_k))));
    }
    double sigma = e / (top - cc);
    const double sigma1 = sigma;
    PMFCHK(apply(Ya, Z));
    hipLaunchKernelGGL(k_topk_cheb, blocks(cnt), dim3(256), 0, c->stream, Z, nl > 0 ? D : nullptr, Ya, (const double*)nullptr, Yb, cnt, cc,
                       sigma1 / e, 0.0);
    HIPCHK(c, hipGetLastError());
    for (int d = 2; d <= deg; ++d) {
      const double sigma2 = 1.0 / (2.0 / sigma1 - sigma);
      PMFCHK(apply(Yb, Z));
      hipLaunchKernelGGL(k_topk_cheb, blocks(cnt), dim3(256), 0, c->stream, Z, nl > 0 ? D : nullptr, Yb, Ya, Yc, cnt, cc, 2.0 * sigma2 / e,
                         sigma * sigma2);
      HIPCHK(c, hipGetLastError());
      double* t = Ya; Ya = Yb; Yb = Yc; Yc = t;
      sigma = sigma2;
    }
    std::swap(Ya, Yb);
    PMFCHK(ortho(Ya));
    PMFCHK(rayleigh_ritz(Ya));
  }
  // out of iterations with pairs still unlocked whose Ritz values are NOT negligible: the solver did not converge (the
  // caller falls back to Jacobi where that exists) -- "fewer than num_bases eigenvalues" would be the wrong diagnosis
  if (nl < k && th[0] > 1e-14 * scale)
    return fail(c, PMF_EHIP, "pmf_nndsvd_init: the top-k eigen-solver did not converge (" + std::to_string(nl) + " of " +
                std::to_string(k) + " pairs in " + std::to_string(kMaxIter) + " filter steps)");
  std::vector<double> out(kp16, -1.0);
  for (int j = 0; j < nl && j < kp16; ++j) out[j] = thl[j];
  HIPCHK(c, hipMemcpyAsync(ev_dev, out.data(), (size_t)kp16 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *nl_out = nl;
  if (products_out) *products_out = products;
  return PMF_OK;
}

int nndsvd_init(pmf_ctx* c, int32_t* rank_found) {
  if (c->v_csr) return fail(c, PMF_EINVAL, "pmf_nndsvd_init: dense V only");
  if (c->n > PMF_TOPK_MAX_N)
    return fail(c, PMF_EINVAL, "pmf_nndsvd_init: num_samples <= " + std::to_string(PMF_TOPK_MAX_N) +
                " (the Gram matrix is n x n; pass the transposed problem for wide data)");
  if (c->k > c->n) return fail(c, PMF_EINVAL, "pmf_nndsvd_init: num_bases exceeds the number of columns");
  const int n = (int)c->n, np = c->np, KP = c->KP, ld = np;
  // all n eigenpairs by Jacobi (pmf_nndsvd.h: exact and quick up to ~1000 columns, 13 s at 4096), or the k largest by
  // filtered subspace iteration (pmf_topk.h: 0.06 s instead of 1.2 s at 1500 columns, the only form beyond 4096).
  // pmf_set_option("nndsvd_topk", 1 / 0) forces one of them where both apply; a top-k solve that stalls falls back
  // to Jacobi where that exists.
  const bool topk_fits = c->k + 16 <= (n / 16) * 16;
  bool topk = (n > PMF_NNDSVD_MAX_N) || (topk_fits && (c->opt_nndsvd_topk == 1 || (c->opt_nndsvd_topk < 0 && n > 1024)));
  if (topk && !topk_fits) return fail(c, PMF_EINVAL, "pmf_nndsvd_init: num_bases too close to the number of columns for this size");
  int nj = n + (n & 1);
  DevTemps tmp;
  double *Ad = nullptr, *Ad2 = nullptr, *evals = nullptr, *QT = nullptr, *sv = nullptr, *part = nullptr, *norms = nullptr;
  float *slab = nullptr, *B = nullptr, *wscale = nullptr;
  int *order = nullptr, *info = nullptr, *wmode = nullptr;
  const int64_t blocks16 = c->mp / 16;
  int gchunks = (int)std::min<int64_t>(512, blocks16);
  const int rpc = (int)((blocks16 + gchunks - 1) / gchunks) * 16;     // (small chunks on purpose: fp32 sums inside a chunk, float64 across)
  gchunks = (int)((c->mp + rpc - 1) / rpc);
  const int kp16 = (int)round_up(c->k, 16);
  const bool can_jacobi = n <= PMF_NNDSVD_MAX_N;
  PMFCHK(talloc(c, tmp, &Ad, (size_t)np * np));
  PMFCHK(talloc(c, tmp, &evals, (size_t)std::max(np, kp16)));
  PMFCHK(talloc(c, tmp, &slab, (size_t)gchunks * 128 * (np + 128)));
  PMFCHK(talloc(c, tmp, &B, (size_t)KP * np));
  PMFCHK(talloc(c, tmp, &sv, (size_t)KP));
  PMFCHK(talloc(c, tmp, &order, (size_t)KP));
  PMFCHK(talloc(c, tmp, &info, 2));
  PMFCHK(talloc(c, tmp, &wscale, (size_t)KP));
  PMFCHK(talloc(c, tmp, &wmode, (size_t)KP));
  const int nblk = (int)std::min<int64_t>(512, (c->m + 255) / 256);
  const int64_t rows_per_blk = (c->m + nblk - 1) / nblk;
  PMFCHK(talloc(c, tmp, &part, (size_t)nblk * 2 * KP));
  PMFCHK(talloc(c, tmp, &norms, (size_t)2 * KP));

  // 1. A = V^T V over all ranks' rows
  PMFCHK(gram_vtv(c, Ad, slab, gchunks, rpc));
  // 2./3. eigen-decomposition, top-k selection
  if (topk) {
    int nl = 0;
    PMFCHK(talloc(c, tmp, &QT, (size_t)kp16 * np));
    DevTemps work;                                   // the solver's block buffers: freed before the rest of the pipeline
    const int trc = eigh_topk(c, work, Ad, n, np, c->k, QT, evals, &nl, &c->nndsvd_products);
    if (trc == PMF_OK) {
      nj = kp16;                                     // evals[nl ..] = -1: below the reference's 1e-8 cut
    } else if (can_jacobi) {
      topk = false;                                  // (the message of the failed solve is replaced by whatever follows)
    } else {
      return trc;
    }
  }
  if (!topk) {
    PMFCHK(talloc(c, tmp, &QT, (size_t)np * np));
    PMFCHK(talloc(c, tmp, &Ad2, (size_t)np * np));
    PMFCHK(jacobi_eigh_dev(c, Ad, Ad2, QT, ld, nj, evals, info + 1));
  }
  hipLaunchKernelGGL(k_nndsvd_select, dim3(1), dim3(1024), 0, c->stream, evals, QT, ld, nj, n, c->k, KP, np, B, sv,
                     order, info);
  HIPCHK(c, hipGetLastError());
  int hinfo[2] = {0, 0};
  HIPCHK(c, hipMemcpyAsync(hinfo, info, sizeof(hinfo), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (rank_found) *rank_found = hinfo[0];
  if (hinfo[0] < c->k)
    return fail(c, PMF_EINVAL, "pmf_nndsvd_init: only " + std::to_string(hinfo[0]) + " eigenvalues of data^T data exceed 1e-8 "
                "(svd.py:130-131), fewer than num_bases (the reference raises IndexError at nndsvd.py:94)");
  // 4. U = V (v_i / s_i)  -> dW
  PMFCHK(rowgemm<EPI_STORE>(c, c->dV, np, np, B, np, nullptr, nullptr, c->dW));
  // 5. split norms over all ranks' rows, closed form
  hipLaunchKernelGGL(k_split_norms, dim3((unsigned)nblk, (unsigned)((KP + 255) / 256)), dim3(256), 0, c->stream, c->dW, c->m, KP,
                     rows_per_blk, part);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(k_split_sum, dim3((unsigned)((2 * KP + 255) / 256)), dim3(256), 0, c->stream, part, nblk, KP, norms);
  HIPCHK(c, hipGetLastError());
  PMFCHK(allreduce_sum(c, norms, (size_t)2 * KP, true));
  hipLaunchKernelGGL(k_nndsvd_finalize, dim3(1), dim3(1024), 0, c->stream, QT, ld, order, sv, norms, n, c->k, KP, np,
                     c->dH, wscale, wmode);
  HIPCHK(c, hipGetLastError());
  const int64_t total = c->mp * KP;
  hipLaunchKernelGGL(k_nndsvd_w, dim3(elem_grid(total)), dim3(256), 0, c->stream, c->dW, total, KP,
                     c->m, wscale, wmode);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_w = c->have_h = true; c->hd_synced = false; c->hd_force = true;
  c->g_valid = c->ps_valid = c->num_valid = c->trace_ready = false; c->g_parts = 0;
  return PMF_OK;
}

// ---- NMF (multiplicative update) ---------------------------------------------------------
// Leaves c->resid_parts float64 partials in c->dPart.
template <int NT, bool RNMF>
int launch_resid_t(pmf_ctx* c, float lamb, const float* V, const float* W, int64_t rows_p) {
  const int ntiles = (int)(rows_p / 64);
  const size_t res_smem = resid_res_smem_bytes<NT>(c->np);
  if (c->opt_resid_resident && res_smem <= 150 * 1024 && ntiles >= 64) {
    // H resident in LDS, persistent workgroups (a fixed count: the partials' grouping must not depend on the part)
    static bool res_attr_dev[PMF_MAX_DEVICES] = {};
    bool& res_attr = res_attr_dev[pmf_current_device()];
    if (!res_attr) {
      HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resid_res<NT, RNMF>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      res_attr = true;
    }
    hipLaunchKernelGGL((k_resid_res<NT, RNMF>), dim3((unsigned)std::min(ntiles, 512)), dim3(256), res_smem, c->stream, V,
                       (int64_t)c->np, c->np, W, c->dH, (int64_t)c->np, lamb, c->dD, c->dPart, ntiles);
    HIPCHK(c, hipGetLastError());
    c->resid_parts = std::min(ntiles, 512);
    return PMF_OK;
  }
  const size_t smem = resid_smem_bytes<NT>();
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resid<NT, RNMF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_done = true;
  }
  hipLaunchKernelGGL((k_resid<NT, RNMF>), dim3((unsigned)ntiles), dim3(256), smem, c->stream, V,
                     (int64_t)c->np, c->np, W, c->dH, (int64_t)c->np, lamb, c->dD, c->dPart);
  HIPCHK(c, hipGetLastError());
  c->resid_parts = ntiles;
  return PMF_OK;
}

int launch_resid(pmf_ctx* c, bool rnmf, float lamb, const float* V = nullptr, const float* W = nullptr,
                 int64_t rows_p = 0) {
  if (!V) { V = c->dV; W = c->dW; rows_p = c->mp; }
  switch (c->NT) {
    case 1: return rnmf ? launch_resid_t<1, true>(c, lamb, V, W, rows_p) : launch_resid_t<1, false>(c, lamb, V, W, rows_p);
    case 2: return rnmf ? launch_resid_t<2, true>(c, lamb, V, W, rows_p) : launch_resid_t<2, false>(c, lamb, V, W, rows_p);
    case 4: return rnmf ? launch_resid_t<4, true>(c, lamb, V, W, rows_p) : launch_resid_t<4, false>(c, lamb, V, W, rows_p);
    case 8: return rnmf ? launch_resid_t<8, true>(c, lamb, V, W, rows_p) : launch_resid_t<8, false>(c, lamb, V, W, rows_p);
  }
  return fail(c, PMF_EINVAL, "bad NT");
}

// num_bases > 128: sum((V - W H)^2) over this rank's rows -> *dst (device), plain-FMA tiles; rnmf: D = S - V too
int resid_bigk(pmf_ctx* c, bool rnmf, double* dst, const float* V = nullptr, const float* W = nullptr, int64_t rows_p = 0) {
  if (!V) { V = c->dV; W = c->dW; rows_p = c->mp; }      // (a row tile of a streamed pass otherwise)
  const int gx = c->np / 64, gy = (int)(rows_p / 64);
  const int nb2 = gx * gy;
  DevTemps tmp;                       // frees `part` on every exit
  double* part = nullptr;
  PMFCHK(talloc(c, tmp, &part, (size_t)nb2));
  if (rnmf)
    hipLaunchKernelGGL(k_resid_bigk<true>, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, c->stream, V, (int64_t)c->np, W,
                       c->KP, c->dH, (int64_t)c->np, part, (float)c->lamb_w, c->dD);
  else
    hipLaunchKernelGGL(k_resid_bigk<false>, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, c->stream, V, (int64_t)c->np, W,
                       c->KP, c->dH, (int64_t)c->np, part, 0.f, (float*)nullptr);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(256), 0, c->stream, part, nb2, dst);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));   // the scratch is freed on return
  return PMF_OK;
}

int rnmf_update_s(pmf_ctx* c) {   // rnmf.py:96-98; also leaves sum((V - W H)^2) in c->rnmf_err2
  const int nb = (int)(c->mp / 64);
  const float lamb = (float)c->lamb_w;
  if (c->nb > 1) {
    PMFCHK(resid_bigk(c, true, c->dScal + 4));
  } else {
    PMFCHK(launch_resid(c, true, lamb));
    hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(256), 0, c->stream, c->dPart, c->resid_parts, c->dScal + 4);
  }
  HIPCHK(c, hipGetLastError());
  PMFCHK(allreduce_sum(c, c->dScal + 4, 1, true));
  HIPCHK(c, hipMemcpyAsync(&c->rnmf_err2, c->dScal + 4, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  c->s_valid = true;
  return PMF_OK;
}


// ---- num_bases > 128 (NMF only): the update rules in blocks of 128 bases on the NT = 8 kernels -------
// W step (nmf.py:128-132): Num = V H^T and Den = W (H H^T) block by block into [mp][KP] buffers, then
// one elementwise pass.  (P | S) (nmf.py:122-124 operands): per base block W_b^T V, and W_b^T W by the
// same kernel with W in the place of V.
// X [rows_p][np] (V, D = S - data, or a streamed tile), Wr / W1r / W2r the same rows of W and of the two [.][KP] temporaries
int bigk_update_w_rows(pmf_ctx* c, const float* X, float* Wr, float* W1r, float* W2r, int64_t rows_p, int64_t mvalid) {
  const bool rn = c->algo == PMF_ALGO_RNMF;
  for (int b = 0; b < c->nb; ++b)                            // Den = W G^T, every block from the OLD W
    PMFCHK((launch_rowgemm<8, EPI_STORE>(c, Wr, c->KP, c->KP, c->dG + (size_t)b * 128 * c->KP, c->KP, nullptr, nullptr,
                                         W2r + b * 128, rows_p, mvalid, c->KP)));
  if (c->opt_rowgemm_stream && c->np % 128 == 0 && c->np <= PMF_WIDE_K) {
    // Num = V H_b^T with the update rule as its epilogue: block b of W is rewritten in place (V H^T does not read W)
    const int ntiles = (int)(rows_p / 32);
    const dim3 grid((unsigned)std::min((ntiles + 3) / 4, 512));     // persistent workgroups (k_rowgemm_stream)
    const size_t smem = (size_t)2 * 128 * 64 * sizeof(float);
    for (int b = 0; b < c->nb; ++b) {
      const float* Hb = c->dH + (size_t)b * 128 * c->np;
      float* Wb = Wr + b * 128;
      const float* Db = W2r + b * 128;
      const int kv = std::max(0, std::min(128, c->k - 128 * b));
      if (rn)
        hipLaunchKernelGGL((k_rowgemm_stream<8, 2, EPI_RNMF_W, true>), grid, dim3(256), smem, c->stream, X, (int64_t)c->np, c->np, Hb,
                           (int64_t)c->np, Wb, Db, (float*)nullptr, (int64_t)0, 0.f, mvalid, kv, ntiles, (int64_t)c->KP);
      else if (c->algo == PMF_ALGO_BNMF)
        hipLaunchKernelGGL((k_rowgemm_stream<8, 2, EPI_BNMF_W, true>), grid, dim3(256), smem, c->stream, X, (int64_t)c->np, c->np, Hb,
                           (int64_t)c->np, Wb, Db, (float*)nullptr, (int64_t)0, (float)c->lamb_w, mvalid, kv, ntiles, (int64_t)c->KP);
      else
        hipLaunchKernelGGL((k_rowgemm_stream<8, 2, EPI_NMF_W, true>), grid, dim3(256), smem, c->stream, X, (int64_t)c->np, c->np, Hb,
                           (int64_t)c->np, Wb, Db, (float*)nullptr, (int64_t)0, 0.f, mvalid, kv, ntiles, (int64_t)c->KP);
      HIPCHK(c, hipGetLastError());
    }
    return PMF_OK;
  }
  PMFCHK(rowgemm<EPI_STORE>(c, X, c->np, c->np, c->dH, c->np, nullptr, nullptr, W1r, rows_p, mvalid));   // (all blocks; in chunks of columns when wide)
  const int64_t count = rows_p * c->KP;
  hipLaunchKernelGGL(k_nmf_w_elem, dim3(elem_grid(count)), dim3(256), 0, c->stream, Wr, W1r, W2r, count,
                     c->algo == PMF_ALGO_BNMF ? 1 : rn ? 2 : 0, (float)c->lamb_w, c->KP, mvalid, c->k);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

int bigk_update_w(pmf_ctx* c) {
  PMFCHK(ensure_gram(c, 0.0));
  const bool rn = c->algo == PMF_ALGO_RNMF;      // rnmf.py:109-115: the contraction runs on D = S - data
  if (rn && !c->s_valid) return fail(c, PMF_EINVAL, "RNMF: S does not exist yet (init_h / update_s create it, rnmf.py:94-98)");
  return bigk_update_w_rows(c, rn ? c->dD : c->dV, c->dW, c->dW1, c->dW2, c->mp, c->m);
}

// acc == nullptr: (P | S) of rows [0, rows_p) of X / Wr into dPS; else added (first: stored) to the float64 image acc
int bigk_ps_rows(pmf_ctx* c, const float* Xv, const float* Wr, int64_t rows_p, int rpc, int nch, double* acc, int first) {
  const int64_t ldp = (int64_t)c->np + c->KP;
  for (int b = 0; b < c->nb; ++b) {
    for (int pass = 0; pass < 2; ++pass) {                 // 0: W_b^T V -> P rows,  1: W_b^T W -> S rows
      const float* X = pass == 0 ? Xv : Wr;
      const int xn = pass == 0 ? c->np : c->KP;
      PMFCHK((launch_colgemm<8, false>(c, X, xn, xn, Wr + b * 128, c->KP, rows_p, rpc, nch)));
      const int64_t cnt4 = (int64_t)128 * xn / 4;
      const size_t off = (size_t)b * 128 * ldp + (pass == 0 ? 0 : c->np);
      if (acc)
        hipLaunchKernelGGL((k_reduce_slabs_block<double>), dim3((unsigned)((cnt4 + 63) / 64)), dim3(1024), 0, c->stream, c->dSlab,
                           nch, 128, xn + 128, xn, acc + off, ldp, first ? 0 : 1);
      else
        hipLaunchKernelGGL((k_reduce_slabs_block<float>), dim3((unsigned)((cnt4 + 63) / 64)), dim3(1024), 0, c->stream, c->dSlab,
                           nch, 128, xn + 128, xn, c->dPS + off, ldp, 0);
      HIPCHK(c, hipGetLastError());
    }
  }
  return PMF_OK;
}

int bigk_ps(pmf_ctx* c) {
  return bigk_ps_rows(c, c->algo == PMF_ALGO_RNMF ? c->dD : c->dV, c->dW, c->mp, c->rows_per_chunk, c->nchunks, nullptr, 0);
}

int nmf_fused_pass(pmf_ctx* c);

int nmf_update_w(pmf_ctx* c) {
  if (c->nb > 1) return bigk_update_w(c);
  // The single hook on a fused-kernel shape runs the same one-pass kernel: W is updated and, for the
  // price of the second half of the pass, (W^T V | W^T W) of the new W is already there when
  // update_h() follows (it then costs one k x n sized kernel) -- 0.65 ms for the pair at cfg4
  // instead of 1.12 ms as two tiled passes.
  if ((c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_BNMF) && c->fused_wgs > 0 && !c->fixed_h_loop && !use_csr(c))
    return nmf_fused_pass(c);
  PMFCHK(ensure_gram(c, 0.0));
  if (c->algo == PMF_ALGO_RNMF && !c->s_valid)
    return fail(c, PMF_EINVAL, "RNMF: S does not exist yet (init_h / update_s create it, rnmf.py:94-98)");
  if (c->np > PMF_WIDE_K)       // more columns than one accumulation chain should span: V H^T in chunks, the rule element-wise
    return wide_update_w_rows(c, c->algo == PMF_ALGO_RNMF ? c->dD : c->dV, c->dW, c->mp, c->m);
  if (c->algo == PMF_ALGO_RNMF)
    return rowgemm<EPI_RNMF_W>(c, c->dD, c->np, c->np, c->dH, c->np, c->dW, c->dG, nullptr);
  if (c->algo == PMF_ALGO_BNMF)
    return rowgemm<EPI_BNMF_W>(c, c->dV, c->np, c->np, c->dH, c->np, c->dW, c->dG, nullptr);
  if (c->fixed_h_loop) {
    // H is not updated in this loop, so Num = V H^T is the same every iteration: the first one
    // stores it, the others read it back and never touch V (W*G and the epilogue are all that is left)
    if (!c->dW1) PMFCHK(dalloc(c, &c->dW1, (size_t)std::max<int64_t>(c->mp, c->np) * c->KP));
    if (c->num_valid)
      return rowgemm<EPI_NMF_W_CACHED>(c, c->dV, c->np, c->np, c->dH, c->np, c->dW, c->dG, c->dW1);
    PMFCHK(rowgemm<EPI_NMF_W_SAVE>(c, c->dV, c->np, c->np, c->dH, c->np, c->dW, c->dG, c->dW1));
    c->num_valid = true;
    return PMF_OK;
  }
  stat_begin(c, SITE_ROWGEMM_W);
  const int wrc = rowgemm<EPI_NMF_W>(c, c->dV, c->np, c->np, c->dH, c->np, c->dW, c->dG, nullptr);
  stat_end(c, SITE_ROWGEMM_W);
  return wrc;
}

template <int NT, bool BNMF, bool FOLD>
int launch_h_gram_as(pmf_ctx* c) {
  constexpr size_t smem = hgram_smem_bytes<NT>();
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nmf_h_gram<NT, BNMF, FOLD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_done = true;
  }
  const int wgs = std::min(c->np / 64, PMF_HGRAM_MAX_WGS);
  // inside pmf_factorize's fused loop the next reader of G is the fused kernel, which adds the
  // per-workgroup partials itself: the kernel then ends without waiting for its last workgroup
  const int final_sum = c->gram_partial_ok ? 0 : 1;
  hipLaunchKernelGGL((k_nmf_h_gram<NT, BNMF, FOLD>), dim3((unsigned)wgs), dim3(1024), smem, c->stream, c->dH, c->np, c->dPS,
                     c->dG, (double*)nullptr /* no reader of the float64 copy on the NMF/BNMF paths */, BNMF ? (float)c->lamb_h : 0.f, c->want_trace ? c->dScal + 2 : nullptr,
                     c->dGpart, c->dT1part, c->dTicket, c->stop_arg, final_sum,
                     FOLD ? c->ipc : IpcPeers{}, c->fold_seq, c->fold_flags, c->dIpcErr, c->ipc_wait_ticks,
                     c->profile ? c->dIpcWait : nullptr);
  c->fold_seq = 0;                    // consumed
  HIPCHK(c, hipGetLastError());
  c->g_parts = final_sum ? 0 : wgs;
  c->trace_parts = final_sum ? 0 : wgs;
  return PMF_OK;
}
// the folded exchange's consumer is an instantiation of its own (FOLD): the one-rank kernel carries none of it
template <int NT, bool BNMF>
int launch_h_gram(pmf_ctx* c) {
  return (c->fold_seq && c->ipc.nranks > 1) ? launch_h_gram_as<NT, BNMF, true>(c) : launch_h_gram_as<NT, BNMF, false>(c);
}

// NMF / BNMF: H step and G = H H^T in one launch.  false: not for this algorithm.
bool nmf_h_gram(pmf_ctx* c, int* rc) {
  if (c->algo != PMF_ALGO_NMF && c->algo != PMF_ALGO_BNMF) return false;   // RNMF: generic k_nmf_h
  if (c->nb > 1) return false;                                              // num_bases > 128: generic k_nmf_h
  const bool b = c->algo == PMF_ALGO_BNMF;
  switch (c->NT) {
    case 1: *rc = b ? launch_h_gram<1, true>(c) : launch_h_gram<1, false>(c); return true;
    case 2: *rc = b ? launch_h_gram<2, true>(c) : launch_h_gram<2, false>(c); return true;
    case 4: *rc = b ? launch_h_gram<4, true>(c) : launch_h_gram<4, false>(c); return true;
    case 8: *rc = b ? launch_h_gram<8, true>(c) : launch_h_gram<8, false>(c); return true;
  }
  return false;
}

template <int NT, int CT>
int launch_snmf_h(pmf_ctx* c) {
  constexpr size_t smem = snmf_h_smem_bytes<NT, CT>();
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_snmf_h_mfma<NT, CT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_done = true;
  }
  hipLaunchKernelGGL((k_snmf_h_mfma<NT, CT>), dim3((unsigned)(c->np / (16 * CT))), dim3(1024), smem, c->stream, c->dH, c->np,
                     c->dPS, c->stop_arg);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

int snmf_h_step(pmf_ctx* c) {   // snmf.py:72-91 on MFMA, one workgroup per 64-column panel
  if (c->nb > 1) {                // num_bases > 128: the generic column-block kernel
    hipLaunchKernelGGL(k_nmf_h, dim3((unsigned)(c->np / 16)), dim3(256), (size_t)c->KP * 16 * sizeof(float), c->stream, c->dH,
                       (int64_t)c->np, c->np, c->KP, c->dPS, 3, 0.f, c->k, (int)c->n);
    return PMF_OK;
  }
  if (snmf_h64(c)) {              // H in float64 (pmf_inv.h: k_snmf_h_f64), P / S in float64 inside the Gram-space loop
    PMFCHK(ensure_hd(c));
    const int64_t ldp = (int64_t)c->np + c->KP;
    const dim3 grid((unsigned)(c->np / 16));
#define PMF_SNMF_H64(NT_)                                                                                                      \
    if (c->ps_f64) hipLaunchKernelGGL((k_snmf_h_f64<NT_, double>), grid, dim3(64 * NT_), 0, c->stream, c->dHd, c->dH, c->np,   \
                                      (const double*)c->dPd, (int64_t)c->np, (const double*)c->dSd, (int64_t)c->KP, c->stop_arg); \
    else hipLaunchKernelGGL((k_snmf_h_f64<NT_, float>), grid, dim3(64 * NT_), 0, c->stream, c->dHd, c->dH, c->np,              \
                            (const float*)c->dPS, ldp, (const float*)c->dPS + c->np, ldp, c->stop_arg)
    switch (c->NT) {
      case 1: PMF_SNMF_H64(1); break;
      case 2: PMF_SNMF_H64(2); break;
      case 4: PMF_SNMF_H64(4); break;
      case 8: PMF_SNMF_H64(8); break;
      default: return fail(c, PMF_EINVAL, "bad NT");
    }
#undef PMF_SNMF_H64
    HIPCHK(c, hipGetLastError());
    return PMF_OK;
  }
  const bool narrow = c->np <= 256;   // few 64-column panels: 16-column workgroups spread the step over more CUs
  switch (c->NT) {
    case 1: return launch_snmf_h<1, 4>(c);
    case 2: return launch_snmf_h<2, 4>(c);
    case 4: return narrow ? launch_snmf_h<4, 1>(c) : launch_snmf_h<4, 4>(c);
    case 8: return narrow ? launch_snmf_h<8, 1>(c) : launch_snmf_h<8, 4>(c);
  }
  return fail(c, PMF_EINVAL, "bad NT");
}

// dPS holds (W^T V | W^T W) of the current W summed over ALL ranks (ps_valid).  It does not depend
// on H, so repeated H steps with an unchanged W -- factorize(compute_w=False), the reference's
// documented "coefficients for an existing basis" use (nmf.py:56-65) -- reuse it: after the first
// iteration such a loop costs one k x n sized kernel per iteration and no pass over V at all.
int h_step_from_ps(pmf_ctx* c) {
  int hrc = PMF_OK;
  if (nmf_h_gram(c, &hrc)) {
    PMFCHK(hrc);
    c->g_valid = true;     // G (pad rows/cols are zero because the padded H rows are zero)
    c->num_valid = false;  // H changed
    c->ps_valid = true;
    c->trace_ready = c->want_trace;
    if (c->algo == PMF_ALGO_BNMF) { c->lamb_w *= 1.1; c->lamb_h *= 1.1; }   // bnmf.py:84-85
    return PMF_OK;
  }
  const size_t smem = (size_t)c->KP * 16 * sizeof(float);
  if (c->algo == PMF_ALGO_SNMF)
    PMFCHK(snmf_h_step(c));
  else
    hipLaunchKernelGGL(k_nmf_h, dim3((unsigned)(c->np / 16)), dim3(256), smem, c->stream, c->dH,
                       (int64_t)c->np, c->np, c->KP, c->dPS,
                       c->algo == PMF_ALGO_BNMF ? 1 : c->algo == PMF_ALGO_RNMF ? 2 : 0, (float)c->lamb_h,
                       c->k, (int)c->n);
  HIPCHK(c, hipGetLastError());
  c->g_valid = false; c->g_parts = 0; c->num_valid = false;
  c->ps_valid = true;    // dPS belongs to the current W (update_h never touches W)
  c->trace_ready = false;
  if (c->algo == PMF_ALGO_BNMF) { c->lamb_w *= 1.1; c->lamb_h *= 1.1; }   // bnmf.py:84-85
  return PMF_OK;
}

int ps_tiled(pmf_ctx* c) {   // dPS = (W^T V | W^T W) over this rank's rows
  if (c->nb > 1) return bigk_ps(c);
  if (use_csr(c)) return csr_ps(c);
  PMFCHK(colgemm(c));
  return reduce_slabs(c, c->nchunks);
}

int ensure_ps(pmf_ctx* c) {  // two-pass path: (re)build the all-rank (P | S) unless it is current
  if (c->ps_valid) return PMF_OK;
  PMFCHK(materialize_w(c));
  PMFCHK(ps_tiled(c));
  PMFCHK(allreduce_ps(c));
  c->ps_valid = true;
  return PMF_OK;
}

int nmf_update_h(pmf_ctx* c) {
  if (c->algo == PMF_ALGO_RNMF) {                // rnmf.py:100-107: H step on D = S - data, then update_s
    if (!c->s_valid) return fail(c, PMF_EINVAL, "RNMF: S does not exist yet (init_h / update_s create it, rnmf.py:94-98)");
    c->ps_valid = false;                         // D changed in the last update_s
    PMFCHK(ensure_ps(c));
    PMFCHK(h_step_from_ps(c));
    c->ps_valid = false;                         // (P | S) were built from D, not from V
    return rnmf_update_s(c);
  }
  PMFCHK(ensure_ps(c));
  return h_step_from_ps(c);
}

// One pass over V doing update_w AND the partials for update_h (pmf_fused.h): W is updated and the
// all-rank (P | S) of the NEW W is left in dPS.
int nmf_fused_pass(pmf_ctx* c) {
  c->ps_valid = false;
  c->trace_ready = false;       // <P,H>, <S,G> belong to the old W
  const float* Gsrc = c->dG;
  int ngp = 0;
  if (c->g_valid && c->g_parts > 0 && !c->fused8) { Gsrc = c->dGpart; ngp = c->g_parts; }   // partial sums, added by the kernel
  else PMFCHK(ensure_gram(c, 0.0));
  const bool rn = c->algo == PMF_ALGO_RNMF;     // rnmf.py:100-115: both contractions run on D = S - data
  if (rn && !c->s_valid) return fail(c, PMF_EINVAL, "RNMF: S does not exist yet (init_h / update_s create it, rnmf.py:94-98)");
  if (c->fused8) {               // the cooperative form (pmf_coop.h)
    stat_begin(c, SITE_FUSED);
    const int lrc8 = pmf_launch_coop(c->stream, rn ? FUSED_RNMF : c->algo == PMF_ALGO_BNMF ? FUSED_BNMF : FUSED_NMF, c->NT, c->np,
                                 rn ? c->dD : c->dV, c->dW, c->dH, c->dG, c->mp, c->fused_wgs, (float)c->lamb_w, c->dSlab,
                                 c->stop_arg);
    stat_end(c, SITE_FUSED);
    if (lrc8 != PMF_OK) return fail(c, lrc8, "cooperative one-pass kernel launch failed");
    HIPCHK(c, hipGetLastError());
    const int NTP8 = c->np / 16, KT8 = c->KP / 16;
    pmf_launch_reduce_slabs_coop(c->stream, c->dSlab, c->fused_wgs, c->coop_bt, NTP8, KT8, c->np, c->dPS, c->stop_arg);
    HIPCHK(c, hipGetLastError());
    PMFCHK(allreduce_ps(c));
    c->ps_valid = true;
    return PMF_OK;
  }
  const FusedCtl ctl = take_fused_ctl(c);
  hipEvent_t se0 = nullptr, se1 = nullptr;
  stat_pair(c, SITE_FUSED, &se0, &se1);          // (profiling: the pair rides on the dispatch itself, no barrier packets in the loop)
  const int lrc = pmf_launch_fused(c->stream, rn ? FUSED_RNMF : c->algo == PMF_ALGO_BNMF ? FUSED_BNMF : FUSED_NMF, c->NT,
                               c->np, rn ? c->dD : c->dV, c->dW, c->dH, Gsrc, c->mp, c->fused_wgs, (float)c->lamb_w,
                               c->dSlab, ctl, ngp, se0, se1);
  if (lrc != PMF_OK) return fail(c, lrc, "fused kernel launch failed");
  HIPCHK(c, hipGetLastError());
  {
    const int NTP = c->np / 16;
    const int ntu = c->NT * NTP + c->NT * (c->NT + 1) / 2;
    // the folded exchange: this launch pushes the rank's partial tiles to every peer, the H-step launch behind it waits for
    // the peers' and adds them in rank order (h_step_from_ps -> launch_h_gram) -- only inside nmf_fused_iteration, where that
    // launch is certain to follow on every rank
    const bool fold = c->fold_loop && c->opt_fold && c->ipc.nranks > 1 && ntu <= PMF_IPC_MAX_WGS && c->np / 64 <= PMF_HGRAM_MAX_WGS &&
                      (size_t)ps_elems(c) * sizeof(float) <= PMF_IPC_MAX_BYTES &&
                      (c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_BNMF) && c->nb == 1;
    const unsigned seq = fold ? ++c->ipc_seq : 0u;
    hipLaunchKernelGGL(k_reduce_slabs_tiles, dim3((unsigned)ntu), dim3(1024), 0, c->stream, c->dSlab,
                       c->fused_wgs, c->NT, NTP, c->np, c->dPS, c->stop_arg, fold ? c->ipc : IpcPeers{}, seq);
    HIPCHK(c, hipGetLastError());
    if (fold) {
      c->fold_seq = seq; c->fold_flags = ntu;
      ++c->ipc_calls; ++c->fold_calls;
      return PMF_OK;                  // dPS becomes the all-rank sum in the prologue of the H-step launch (ps_valid is set there)
    }
  }
  PMFCHK(allreduce_ps(c));
  c->ps_valid = true;
  return PMF_OK;
}

int nmf_fused_iteration(pmf_ctx* c) {
  c->fold_loop = true;
  const int prc = nmf_fused_pass(c);
  c->fold_loop = false;
  PMFCHK(prc);
  PMFCHK(h_step_from_ps(c));
  if (c->algo == PMF_ALGO_RNMF) {               // rnmf.py:107: update_h ends with update_s
    c->ps_valid = false;                        // (P | S) were built from D, not from V
    return rnmf_update_s(c);
  }
  return PMF_OK;
}

// ---- SNMF -----------------------------------------------------------------------------------
// inv(H H^T) in float64 (Gauss-Jordan in registers, identity on the padding), then M^T = inv(H H^T) H in
// float64, rounded once: dMT [KP][np] for the dense kernels, dW1 = M [np][KP] for the CSR kernels.
// snmf.py:69: np.linalg.inv raises LinAlgError("Singular matrix") on a zero pivot; the inverse kernels raise
// dSing instead, read back wherever the host synchronises anyway (end of pmf_update_w / pmf_factorize / a streamed pass).
int check_singular(pmf_ctx* c) {
  if (c->algo != PMF_ALGO_SNMF || !c->dSing) return PMF_OK;
  int flag = 0;
  HIPCHK(c, hipMemcpyAsync(&flag, c->dSing, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (!flag) return PMF_OK;
  HIPCHK(c, hipMemsetAsync(c->dSing, 0, sizeof(int), c->stream));
  return fail(c, PMF_ESINGULAR, "SNMF: H H^T is singular (the reference's np.linalg.inv raises LinAlgError, snmf.py:69)");
}

int launch_inverse(pmf_ctx* c) {   // dGinvD = inv(dGd), float64
  if (!c->dSing) PMFCHK(dalloc(c, &c->dSing, 1));
  if (c->KP <= 64) {                       // blocked Gauss-Jordan on the float64 MFMA (pmf_inv.h)
    hipLaunchKernelGGL((k_inverse_spd_mfma<4>), dim3(1), dim3(256), 0, c->stream, c->dGd, c->KP, c->k, c->dGinvD, c->stop_arg, c->dSing);
  } else if (c->KP <= 128) {
    hipLaunchKernelGGL((k_inverse_spd_mfma<8>), dim3(1), dim3(1024), 0, c->stream, c->dGd, c->KP, c->k, c->dGinvD, c->stop_arg, c->dSing);
  } else {                         // num_bases > 128: the matrix in L2, a cooperative grid (k_inverse_spd_big)
    const size_t E = (size_t)c->KP * c->KP;
    if (!c->dInvA) { PMFCHK(dalloc(c, &c->dInvA, E)); PMFCHK(dalloc(c, &c->dInvB, E)); }
    // dGd stays intact (g_valid covers it): the elimination runs on a copy; a stopped free-running loop keeps dGinvD
    HIPCHK(c, hipMemcpyAsync(c->dInvA, c->dGd, E * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    const unsigned wgs = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cus, (int64_t)E / 4096));
    double *a_ = c->dInvA, *b_ = c->dInvB, *o_ = c->dGinvD;
    int kp_ = c->KP, k_ = c->k;
    const int* stop_ = c->stop_arg;
    int* sing_ = c->dSing;
    void* args[] = {&a_, &b_, &kp_, &k_, &o_, &stop_, &sing_};
    HIPCHK(c, hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_inverse_spd_big), dim3(wgs), dim3(1024), args, 0,
                                         c->stream));
  }
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

int snmf_inverse(pmf_ctx* c) {
  PMFCHK(ensure_gram(c, 1.0));
  PMFCHK(launch_inverse(c));
  if (snmf_h64(c)) {
    PMFCHK(ensure_hd(c));
    hipLaunchKernelGGL(k_snmf_mt<double>, dim3((unsigned)(c->np / 16), (unsigned)(c->KP / 16)), dim3(64), 0, c->stream, c->dHd,
                       (int64_t)c->np, c->np, c->KP, c->dGinvD, use_csr(c) ? (float*)nullptr : c->dMT,
                       use_csr(c) ? c->dW1 : (float*)nullptr, (double*)nullptr, (const int*)nullptr);
  } else {
    hipLaunchKernelGGL(k_snmf_mt<float>, dim3((unsigned)(c->np / 16), (unsigned)(c->KP / 16)), dim3(64), 0, c->stream, c->dH,
                       (int64_t)c->np, c->np, c->KP, c->dGinvD, use_csr(c) ? (float*)nullptr : c->dMT,
                       use_csr(c) ? c->dW1 : (float*)nullptr, (double*)nullptr, (const int*)nullptr);
  }
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

int snmf_fused_pass(pmf_ctx* c);

int snmf_update_w(pmf_ctx* c) {
  if (c->fused_wgs > 0 && !use_csr(c)) return snmf_fused_pass(c);   // as nmf_update_w: one pass, (P | S) kept for update_h
  PMFCHK(snmf_inverse(c));
  if (use_csr(c)) return csr_w(c);
  return rowgemm<EPI_STORE>(c, c->dV, c->np, c->np, c->dMT, c->np, nullptr, nullptr, c->dW);   // W = V M^T
}

int snmf_inverse(pmf_ctx* c);

// SNMF: update_w and the partials of update_h in ONE pass over V (dense data, fused shapes).
int snmf_fused_pass(pmf_ctx* c) {
  c->ps_valid = false;
  c->trace_ready = false;
  PMFCHK(snmf_inverse(c));
  const FusedCtl ctl = take_fused_ctl(c);
  stat_begin(c, SITE_FUSED);
  const int lrc = pmf_launch_fused(c->stream, FUSED_SNMF, c->NT, c->np, c->dV, c->dW, c->dMT, nullptr, c->mp,
                                   c->fused_wgs, 0.f, c->dSlab, ctl, 0);
  stat_end(c, SITE_FUSED);
  if (lrc != PMF_OK) return fail(c, lrc, "fused SNMF kernel launch failed");
  HIPCHK(c, hipGetLastError());
  {
    const int NTP = c->np / 16;
    const int ntu = c->NT * NTP + c->NT * (c->NT + 1) / 2;
    hipLaunchKernelGGL(k_reduce_slabs_tiles, dim3((unsigned)ntu), dim3(1024), 0, c->stream, c->dSlab,
                       c->fused_wgs, c->NT, NTP, c->np, c->dPS, c->stop_arg, IpcPeers{}, 0u);
    HIPCHK(c, hipGetLastError());
  }
  PMFCHK(allreduce_ps(c));
  c->ps_valid = true;
  return PMF_OK;
}

int snmf_fused_iteration(pmf_ctx* c) {
  PMFCHK(snmf_fused_pass(c));
  return h_step_from_ps(c);
}

// ---- SNMF in Gram space ---------------------------------------------------------------------------
// snmf.py:67-70 makes W a LINEAR function of the data once H is given: W = V M, M = H^T inv(H H^T).
// Everything update_h (snmf.py:72-91) takes from W are XW = V^T W and WW = W^T W, i.e.
//     P = W^T V = M^T (V^T V) = M^T C,      S = W^T W = M^T C M = P M,      C = V^T V  (n x n),
// and C does not change during factorize().  So a loop that runs update_w AND update_h needs ONE pass
// over V (C, float64, all-reduced once across the ranks) and then iterates on k x n sized data only:
// G = H H^T -> inv -> M^T (all float64) -> P = M^T C -> S = P M -> the H step -> the error through the
// trace identity (same P, S).  W is materialised once, after the last iteration (W = V M with the M of
// that iteration: exactly the W the reference holds then).  No per-iteration pass over V or W, no
// per-iteration collective; results agree with the pass-per-iteration form to rounding (P, S now come
// out of float64 arithmetic).  CSR data: C by k_csr_gram (pmf_csr.h), dense data: gram_vtv.
int ensure_vgram(pmf_ctx* c) {
  if (c->c_valid) return PMF_OK;
  const int np = c->np;
  if (!c->dC) PMFCHK(dalloc(c, &c->dC, (size_t)np * np));
  if (!c->dMTd) PMFCHK(dalloc(c, &c->dMTd, (size_t)c->KP * np));
  if (!c->dPd) PMFCHK(dalloc(c, &c->dPd, (size_t)c->KP * np));
  if (use_csr(c)) {                 // k_csr_gram: per-workgroup images of C in exact fixed point (pmf_csr.h), added up as integers
    const size_t E = (size_t)np * np;
    const size_t T2 = 2 * gram_tri(np);                                // two 64-bit limbs per entry of the upper triangle
    const int use_lds = T2 * sizeof(unsigned long long) + gram_stage_bytes() <= 160 * 1024;
    const int wgs = use_lds ? 256 : 32;            // global images are T2 words each: fewer of them
    // the grids of the two limbs from the largest |v|: |v| < 2^e  ->  u1 = 2^(2e-32), u2 = 2^(2e-64)
    if (!c->dVmaxBits) PMFCHK(dalloc(c, &c->dVmaxBits, (size_t)1));
    unsigned* mxbits = c->dVmaxBits;
    HIPCHK(c, hipMemsetAsync(mxbits, 0, sizeof(unsigned), c->stream));
    const int64_t nnz = c->nnz;
    if (nnz > 0) {
      hipLaunchKernelGGL(k_absmax_bits_f32, dim3((unsigned)std::min<int64_t>((nnz + 255) / 256, 2048)), dim3(256), 0, c->stream, c->dVals, nnz, mxbits);
      HIPCHK(c, hipGetLastError());
    }
    unsigned hb = 0;
    HIPCHK(c, hipMemcpyAsync(&hb, mxbits, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    float vmax;
    std::memcpy(&vmax, &hb, sizeof(float));
    const int finite = std::isfinite(vmax) ? 1 : 0;
    int ex = 0;
    if (finite && vmax > 0.f) (void)std::frexp(vmax, &ex);             // vmax = f 2^ex, f in [0.5, 1): |v| < 2^ex
    GramScale gs;
    gs.u1 = std::ldexp(1.0, 2 * ex - 32); gs.inv_u1 = std::ldexp(1.0, 32 - 2 * ex); gs.inv_u2 = std::ldexp(1.0, 64 - 2 * ex);
    // per-workgroup images of C: kept with the context (34 MiB at n = 128); zeroed only where the kernel adds
    // into them directly (LDS images are written out whole)
    if (!c->dCslabs) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->dCslabs), (size_t)wgs * T2 * sizeof(unsigned long long)));
    unsigned long long* slabs = reinterpret_cast<unsigned long long*>(c->dCslabs);
    if (!use_lds) HIPCHK(c, hipMemsetAsync(slabs, 0, (size_t)wgs * T2 * sizeof(unsigned long long), c->stream));
    const size_t smem = (use_lds ? T2 * sizeof(unsigned long long) : 0) + gram_stage_bytes();
    static bool attr_done_dev[PMF_MAX_DEVICES] = {};
    bool& attr_done = attr_done_dev[pmf_current_device()];
    if (!attr_done) {
      HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_csr_gram), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(160 * 1024)));
      attr_done = true;
    }
    hipLaunchKernelGGL(k_csr_gram, dim3((unsigned)wgs), dim3(64 * GRAM_WAVES), smem, c->stream, c->dIndptr,
                       c->dIndices, c->dVals, c->m, np, slabs, use_lds, gs);
    HIPCHK(c, hipGetLastError());
    hipLaunchKernelGGL(k_csr_gram_sum, dim3((unsigned)((E + 63) / 64)), dim3(256), 0, c->stream, slabs, wgs, np, c->dC, gs, finite);
    HIPCHK(c, hipGetLastError());
    PMFCHK(allreduce_sum(c, c->dC, E, true));
  } else {
    DevTemps tmp;
    const int64_t blocks16 = c->mp / 16;
    int gchunks = (int)std::min<int64_t>(512, blocks16);
    const int rpc = (int)((blocks16 + gchunks - 1) / gchunks) * 16;   // (small chunks on purpose: fp32 sums inside a chunk, float64 across)
    gchunks = (int)((c->mp + rpc - 1) / rpc);
    float* slab = nullptr;
    PMFCHK(talloc(c, tmp, &slab, (size_t)gchunks * 128 * (np + 128)));
    PMFCHK(gram_vtv(c, c->dC, slab, gchunks, rpc));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // the scratch is freed on return
  }
  c->c_valid = true;
  return PMF_OK;
}

constexpr int PMF_GRAM_MAX_NP = 1024;   // C is np x np float64 (8 MiB at the limit)

// Worth it?  CSR data: always (C costs a few ms on the host).  Dense data: forming C is 2 m n^2 flop, a
// pass-per-iteration step 4 m n k: from about n / 2k iterations on (or when C is already there).
bool snmf_gram_ok(const pmf_ctx* c, int niter) {
  if (c->algo != PMF_ALGO_SNMF || c->np > PMF_GRAM_MAX_NP) return false;
  if (c->opt_snmf_gram == 0) return false;
  if (use_csr(c)) return true;
  return c->opt_snmf_gram >= 1 || c->c_valid || (int64_t)2 * c->k * niter >= c->n;
}

int snmf_gram_iteration(pmf_ctx* c) {
  const int np = c->np, KP = c->KP;
  const int64_t ldp = (int64_t)np + KP;
  c->ps_valid = false;
  c->trace_ready = false;
  PMFCHK(ensure_gram(c, 1.0));
  PMFCHK(launch_inverse(c));
  const bool pipe = w_pipe_on(c);
  float* mcsr = c->dW1;
  if (pipe) {                  // M of this iteration goes into the buffer the write before last has finished reading
    PMFCHK(w_pipe_init(c));
    const int b = (int)(c->w_pipe_it & 1);
    if (c->ev_w_pending[b]) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_w[b], 0)); c->ev_w_pending[b] = false; }
    mcsr = w_pipe_mbuf(c, c->w_pipe_it);
  }
  const bool h64 = snmf_h64(c);
  if (h64) {
    PMFCHK(ensure_hd(c));
    hipLaunchKernelGGL(k_snmf_mt<double>, dim3((unsigned)(np / 16), (unsigned)(KP / 16)), dim3(64), 0, c->stream, c->dHd, (int64_t)np, np, KP,
                       c->dGinvD, use_csr(c) ? (float*)nullptr : c->dMT, use_csr(c) ? mcsr : (float*)nullptr, c->dMTd, c->stop_arg);
  } else {
    hipLaunchKernelGGL(k_snmf_mt<float>, dim3((unsigned)(np / 16), (unsigned)(KP / 16)), dim3(64), 0, c->stream, c->dH, (int64_t)np, np, KP,
                       c->dGinvD, use_csr(c) ? (float*)nullptr : c->dMT, use_csr(c) ? mcsr : (float*)nullptr, c->dMTd, c->stop_arg);
  }
  HIPCHK(c, hipGetLastError());
  if (pipe) {                  // W = V M on the side stream, beside everything that follows here (nothing below reads W)
    const int b = (int)(c->w_pipe_it & 1);
    HIPCHK(c, hipEventRecord(c->ev_mt[b], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->w_stream, c->ev_mt[b], 0));
    // the launch's own HIP events, on the stream it runs on -- sampled like every timed site (profile_every): the pair sits
    // BETWEEN two writes of a loop that is bound by exactly these writes
    const bool timed_w = c->profile && c->stat.site == SITE_MATERIALIZE && (c->stat.seen++ % c->stat.every == 0);
    if (timed_w) {
      KernelStat& st = c->stat;
      if (st.used + 2 > st.ev.size()) for (int q = 0; q < 2; ++q) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) st.ev.push_back(e); }
      if (st.used + 2 <= st.ev.size()) (void)hipEventRecord(st.ev[st.used], c->w_stream);
    }
    PMFCHK(csr_w(c, c->w_stream, mcsr, c->opt_w_pipe));
    if (timed_w && c->stat.used + 2 <= c->stat.ev.size()) {
      (void)hipEventRecord(c->stat.ev[c->stat.used + 1], c->w_stream);
      c->stat.used += 2;
    }
    HIPCHK(c, hipEventRecord(c->ev_w[b], c->w_stream));
    c->ev_w_pending[b] = true;
    ++c->w_pipe_it;
  }
  // P = M^T C  (KP x np), float64 kept for S, float32 into (P | S)
  hipLaunchKernelGGL((k_dgemm_mfma<false>), dim3((unsigned)(np / 16), (unsigned)(KP / 16)), dim3(64), 0, c->stream, c->dMTd,
                     (int64_t)np, c->dC, (int64_t)np, np, c->dPd, (int64_t)np, c->dPS, ldp, c->stop_arg);
  HIPCHK(c, hipGetLastError());
  // S = P M = P (M^T)^T  (KP x KP)
  hipLaunchKernelGGL((k_dgemm_mfma<true>), dim3((unsigned)(KP / 16), (unsigned)(KP / 16)), dim3(64), 0, c->stream, c->dPd,
                     (int64_t)np, c->dMTd, (int64_t)np, np, h64 ? c->dSd : (double*)nullptr, (int64_t)KP, c->dPS + np, ldp, c->stop_arg);
  HIPCHK(c, hipGetLastError());
  c->w_implicit = !pipe;      // dW is stale from here on: W = V M with the M just formed (pipelined: being written already)
  c->ps_valid = true;         // (P | S) of that W, all ranks (C is all-reduced)
  if (c->opt_snmf_gram == 2 && !pipe) PMFCHK(materialize_w(c));   // W rewritten in every iteration, as the reference's update_w does
  c->ps_valid = true;
  c->ps_f64 = h64;            // the H step takes P and S in float64 (dPd, dSd), not their float32 roundings in (P | S)
  const int hrc = h_step_from_ps(c);
  c->ps_f64 = false;
  c->psd_fresh = h64 && hrc == PMF_OK;     // the error of this iteration takes <P,H>, <S H,H> from the float64 P, S and H
  return hrc;
}

// W = V M for the M the last Gram-space iteration formed (dMT dense / dW1 CSR).
int materialize_w(pmf_ctx* c) {
  PMFCHK(w_pipe_join(c));     // (a pipelined write still in flight on the side stream)
  if (!c->w_implicit) return PMF_OK;
  c->w_implicit = false;
  stat_begin(c, SITE_MATERIALIZE);
  int rc = PMF_OK;
  if (use_csr(c)) {
    const bool keep_ps = c->ps_valid;
    rc = csr_w(c);
    c->ps_valid = keep_ps;
  } else {
    rc = rowgemm<EPI_STORE>(c, c->dV, c->np, c->np, c->dMT, c->np, nullptr, nullptr, c->dW);
  }
  stat_end(c, SITE_MATERIALIZE);
  return rc;
}

// CSR SNMF: update_w and the (P | S) partials of update_h in one pass over the CSR rows.
template <int NT>
int launch_csr_fused(pmf_ctx* c, int wgs) {
  const size_t smem = ((size_t)2 * c->np * c->KP + 4 * 16 * c->KP) * sizeof(float);
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_snmf_csr_fused<NT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const int nblk = (int)(c->mp / 16), nw = wgs * 4;
  hipLaunchKernelGGL((k_snmf_csr_fused<NT>), dim3(wgs), dim3(256), smem, c->stream, c->dIndptr, c->dIndices,
                     c->dVals, nblk / nw, nblk % nw, c->np, c->dW1, c->dW, c->dSlab);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

template <int NT, int NTP>
int launch_csr_mfma(pmf_ctx* c, int wgs) {
  const size_t smem = ((size_t)16 * NTP * 16 * NT + 64 * 16 * NTP) * sizeof(float) + 16;
  static bool attr_done_dev[PMF_MAX_DEVICES] = {};   // the attribute is per device
  bool& attr_done = attr_done_dev[pmf_current_device()];
  if (!attr_done) {
    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_snmf_csr_mfma<NT, NTP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_done = true;
  }
  const int nblk = (int)(c->mp / 16), nw = wgs * 4;
  hipLaunchKernelGGL((k_snmf_csr_mfma<NT, NTP>), dim3(wgs), dim3(256), smem, c->stream, c->dIndptr,
                     c->dIndices, c->dVals, nblk / nw, nblk % nw, c->dW1, c->dW, c->dSlab);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

// MFMA variant for the register-resident P shapes; false: not covered (LDS-atomic kernel instead)
bool csr_mfma(pmf_ctx* c, int wgs, int* rc) {
  if (c->np % 16) return false;
  const int key = c->NT * 100 + c->np / 16;
  switch (key) {
    case 808: *rc = launch_csr_mfma<8, 8>(c, wgs); return true;
    case 408: *rc = launch_csr_mfma<4, 8>(c, wgs); return true;
    case 404: *rc = launch_csr_mfma<4, 4>(c, wgs); return true;
    case 208: *rc = launch_csr_mfma<2, 8>(c, wgs); return true;
    case 108: *rc = launch_csr_mfma<1, 8>(c, wgs); return true;
    case 104: *rc = launch_csr_mfma<1, 4>(c, wgs); return true;
  }
  return false;
}

bool csr_fused_ok(const pmf_ctx* c) {
  const size_t smem = ((size_t)2 * c->np * c->KP + 4 * 16 * c->KP) * sizeof(float);
  return use_csr(c) && smem <= 160 * 1024;
}

int snmf_csr_fused_iteration(pmf_ctx* c) {
  c->ps_valid = false;
  PMFCHK(snmf_inverse(c));            // leaves M = H^T inv(H H^T) in dW1
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  int wgs = (int)std::min<int64_t>((c->mp / 16 + 3) / 4, cus);
  wgs = std::min(wgs, c->nchunks > 0 ? std::max(c->nchunks, 1) : wgs);   // slab capacity
  stat_begin(c, SITE_CSR_PASS);
  int mrc = PMF_OK;
  if (csr_mfma(c, wgs, &mrc)) {
    stat_end(c, SITE_CSR_PASS);
    PMFCHK(mrc);
    PMFCHK(reduce_slabs(c, wgs));
    PMFCHK(allreduce_ps(c));
    c->ps_valid = true;
    return h_step_from_ps(c);
  }
  switch (c->NT) {
    case 1: PMFCHK(launch_csr_fused<1>(c, wgs)); break;
    case 2: PMFCHK(launch_csr_fused<2>(c, wgs)); break;
    case 4: PMFCHK(launch_csr_fused<4>(c, wgs)); break;
    case 8: PMFCHK(launch_csr_fused<8>(c, wgs)); break;
    default: return fail(c, PMF_EINVAL, "bad NT");
  }
  stat_end(c, SITE_CSR_PASS);
  PMFCHK(reduce_slabs(c, wgs));
  PMFCHK(allreduce_ps(c));
  c->ps_valid = true;
  return h_step_from_ps(c);
}

int snmf_update_h(pmf_ctx* c) {
  PMFCHK(ensure_ps(c));
  return h_step_from_ps(c);
}

// ---- NMFALS ---------------------------------------------------------------------------------
int nnqp_warm_flag(pmf_ctx* c, hipStream_t s) {   // dWarm[0] = 1 iff the QPs over the current dGd have unique minimisers
  if (!c->dWarm) PMFCHK(dalloc(c, &c->dWarm, 1));
  if (c->k <= 64) {
    // the blocked Gauss-Jordan of k_inverse_spd_mfma meets exactly the pivots of the unpivoted LDL^T, as ratios to the diagonal
    // already (unit-diagonal scaling), dead bases patched out: its `spd_flag` IS the uniqueness test -- 17 us where the
    // one-wave elimination of k_spd_unique (rounds 2-3) took 28; the inverse itself is a by-product nobody reads here
    if (!c->dBinv) PMFCHK(dalloc(c, &c->dBinv, (size_t)2 * c->KP * c->KP));
    hipLaunchKernelGGL((k_inverse_spd_mfma<4>), dim3(1), dim3(256), 0, s, c->dGd, c->KP, c->k, c->dBinv, (const int*)nullptr, (int*)nullptr, c->dWarm,
                       c->dBinv + (size_t)c->KP * c->KP);
  } else {
    if (!c->dInvA) PMFCHK(dalloc(c, &c->dInvA, (size_t)c->KP * c->KP));
    pmf_launch_spd_unique_big(s, c->dGd, c->KP, c->k, c->dInvA, c->dWarm);
  }
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

// (a few thousand problems do not fill the chip four to a wave: the H half step of a tall matrix stays on k_nnqp)
bool nnqp_use_quad(const pmf_ctx* c, int64_t nprob) {
  return c->opt_nnqp_quad && c->k <= 64 && (nprob >= 16384 || c->opt_nnqp_quad == 2);
}
// 64 < num_bases <= 128: k_nnqp_wave (pmf_nnls_wave.h) on B = inv(HA), whatever the number of problems
bool nnqp_use_wave(const pmf_ctx* c) { return c->opt_nnqp_wave && c->k > 64 && c->k <= 128; }

// What a half step's QPs need from HA = dGd alone, on stream s: the uniqueness flag and, for k_nnqp_quad,
// B = inv(HA with its dead variables patched out), one k x k sized launch.
int nnqp_prepare(pmf_ctx* c, hipStream_t s, bool quad) {
  if (quad) {
    // the inverse's own pivots are the uniqueness test (k_inverse_spd_mfma's spd_flag): no k_spd_unique launch
    if (!c->dWarm) PMFCHK(dalloc(c, &c->dWarm, 1));
    if (!c->dBinv) PMFCHK(dalloc(c, &c->dBinv, (size_t)2 * c->KP * c->KP));     // B, and HA with dead variables patched out
    // (dead bases are patched out by the inverse kernel itself, which also writes the patched HA: one launch, not two)
    double* Hp = c->dBinv + (size_t)c->KP * c->KP;
    if (c->k <= 64) hipLaunchKernelGGL((k_inverse_spd_mfma<4>), dim3(1), dim3(256), 0, s, c->dGd, c->KP, c->k, c->dBinv, (const int*)nullptr, (int*)nullptr, c->dWarm, Hp);
    else hipLaunchKernelGGL((k_inverse_spd_mfma<8>), dim3(1), dim3(1024), 0, s, c->dGd, c->KP, c->k, c->dBinv, (const int*)nullptr, (int*)nullptr, c->dWarm, Hp);
    HIPCHK(c, hipGetLastError());
    return PMF_OK;
  }
  return nnqp_warm_flag(c, s);
}

// num_bases > 64: k_nnqp_big keeps one inverse image per workgroup in global memory
int nnqp_scratch(pmf_ctx* c, double** out) {
  *out = nullptr;
  if (c->k <= 64) return PMF_OK;
  if (!c->dQp) {
    const int64_t ks = 64 * pmf_nnqp_big_vpl(c->k);
    PMFCHK(dalloc(c, &c->dQp, (size_t)(pmf_nnqp_big_blocks(c->k, std::max<int64_t>(c->m, c->n)) * ks * ks)));
  }
  *out = c->dQp;
  return PMF_OK;
}

// One half step's problems: F(var, prob) = F[var * f_sk + prob * f_sp], X likewise; HA in dGd.  32 < num_bases <= 64 with a
// well-conditioned HA (dWarm, k_spd_unique): k_nnqp_quad on B = inv(HA); otherwise (and as the fallback the flag
// selects on the device, without a host round trip) k_nnqp / k_nnqp_big.
int solve_nnqps(pmf_ctx* c, const float* F, int64_t f_sk, int64_t f_sp, float* X, int64_t x_sk, int64_t x_sp, int64_t nprob, bool stat,
                bool prepared = false) {
  const bool quad = nnqp_use_quad(c, nprob), wave = nnqp_use_wave(c);
  if (!prepared) PMFCHK(nnqp_prepare(c, c->stream, quad || wave));
  double* qp = nullptr;
  PMFCHK(nnqp_scratch(c, &qp));
  if (stat) stat_begin(c, SITE_NNQP_W);
  int rc = PMF_OK;
  QuadCtl ctl{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  const bool frames = quad && c->opt_nnqp_frame16;
  if (frames) {
    // The 16-slot frame pays when most problems fit it (settled active sets: three waves per SIMD instead of two).  Whether
    // they do is decided ON THE DEVICE from the count the previous half step of this kind left (no host round trip: the
    // loop is enqueued ahead of the GPU).  Counters rotate over the calls and are zeroed by the kernels themselves.
    const int site = stat ? 0 : 1;                   // W / H half step
    if (!c->dNbig) {
      PMFCHK(dalloc(c, &c->dNbig, 10));
      // no history yet: the first half step of either kind goes to the 32-slot frame (from a random start every system is
      // beyond 16 unknowns, and 262 144 problems appending themselves to the list one atomic each is the slowest way to find out)
      for (int st = 0; st < 2; ++st) HIPCHK(c, hipMemsetAsync(c->dNbig + 5 * st + 2, 0x3f, sizeof(int), c->stream));
    }
    if (c->defer_cap < nprob) {
      if (c->dDefer) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->dDefer)); c->dDefer = nullptr; }
      PMFCHK(dalloc(c, &c->dDefer, (size_t)nprob));
      c->defer_cap = nprob;
    }
    int* base = c->dNbig + 5 * site;
    const int64_t t = c->quad_calls[site]++;
    ctl.dlist = c->dDefer;
    ctl.nbig = base + (int)(t % 3);
    ctl.nbig_prev = base + (int)((t + 2) % 3);
    ctl.nbig_next = base + (int)((t + 1) % 3);
    ctl.dcount = base + 3 + (int)(t & 1);
    ctl.dcount_next = base + 3 + (int)((t + 1) & 1);
    if (stat && c->opt_nnqp_count) {                  // the W half step's live counts (counting instantiations)
      if (!c->dQstat) PMFCHK(dalloc(c, &c->dQstat, 8));
      ctl.stats = c->dQstat;
    }
  }
  if (quad) rc = pmf_launch_nnqp_quad(c->stream, c->KP, c->k, c->dGd, c->dBinv + (size_t)c->KP * c->KP, c->dBinv, F, f_sk, f_sp, X, x_sk, x_sp, nprob, c->dWarm,
                                  frames ? &ctl : nullptr, stat && c->opt_nnqp_count != 0);
  if (wave) {
    if (c->y0_cap < nprob) {                         // y0 = inv(HA) f of every problem (k_nnqp_y0): [nprob][KP] float64
      if (c->dY0) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->dY0)); c->dY0 = nullptr; }
      PMFCHK(dalloc(c, &c->dY0, (size_t)nprob * c->KP));
      c->y0_cap = nprob;
    }
    rc = pmf_launch_nnqp_wave(c->stream, c->KP, c->k, c->dGd, c->dBinv + (size_t)c->KP * c->KP, c->dBinv, F, f_sk, f_sp, X, x_sk, x_sp, nprob, c->dWarm, c->dY0);
  }
  if (rc == PMF_OK) rc = pmf_launch_nnqp(c->stream, c->KP, c->k, c->dGd, F, f_sk, f_sp, X, x_sk, x_sp, nprob, c->dWarm, qp, (quad || wave) ? 1 : 0);
  if (stat) stat_end(c, SITE_NNQP_W);
  if (rc != PMF_OK) return fail(c, rc, "nnqp launch failed");
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

int als_update_w(pmf_ctx* c) {
  // HA = H H^T (nmfals.py:93), -FA = V H^T (nmfals.py:88), one QP per row (nmfals.py:89-90)
  bool prepared = false;
  const int ks = c->np >= 2048 && c->np % 512 == 0 ? 8 : c->np >= 512 && c->np % 256 == 0 ? 4 : 1;   // (ensure_gram's rule)
  if (!c->g_valid && c->KP == 64 && c->nb == 1 && ks > 1 && (c->opt_fuse_chain & 1)) {
    // round 6: H H^T split over the columns AND its inverse (flag, patched Hessian, B) in one launch -- the workgroup that
    // completes the last tile inverts (pmf_inv.h: k_gram_splitk<float, true>)
    if (!c->dGramPart) PMFCHK(dalloc(c, &c->dGramPart, (size_t)8 * c->KP * c->KP));
    if (!c->dGramTickets) PMFCHK(dalloc(c, &c->dGramTickets, (size_t)(c->KP / 16) * (c->KP / 16) + 2));
    if (!c->dWarm) PMFCHK(dalloc(c, &c->dWarm, 1));
    if (!c->dBinv) PMFCHK(dalloc(c, &c->dBinv, (size_t)2 * c->KP * c->KP));
    hipLaunchKernelGGL((k_gram_splitk<float, true>), dim3((unsigned)(c->KP / 16), (unsigned)(c->KP / 16), (unsigned)ks), dim3(256), 0, c->stream, c->dH,
                       (int64_t)c->np, c->np, c->KP, c->k, 1.0, c->dG, c->dGd, c->dGramPart, c->dGramTickets, c->dBinv, c->dWarm,
                       c->dBinv + (size_t)c->KP * c->KP);
    HIPCHK(c, hipGetLastError());
    c->g_valid = true; c->g_parts = 0;
    prepared = true;
  } else {
    PMFCHK(ensure_gram(c, 1.0));
  }
  // (The QPs' preparation -- 56 us of single-workgroup k x k kernels that read HA only -- on a second stream beside
  // V H^T was tried: the iteration got 4 % SLOWER, profiles/r03_experiments.md.)
  PMFCHK(rowgemm<EPI_STORE>(c, c->dV, c->np, c->np, c->dH, c->np, nullptr, nullptr, c->dW1));
  return solve_nnqps(c, c->dW1, 1, c->KP, c->dW, 1, c->KP, c->m, true, prepared);
}

int als_update_h(pmf_ctx* c) {
  // HA = W^T W (nmfals.py:78), -FA = W^T V (nmfals.py:73), one QP per column (nmfals.py:74-75)
  c->want_hess = c->nb == 1 && !use_csr(c);
  c->want_inv = true;          // (64 bases, one rank: the slab reduce's last workgroup also inverts the Hessian it completes)
  c->gd_is_s = false;
  c->chain_prepared = false;
  const int prc = ensure_ps(c);
  c->want_hess = false; c->want_inv = false;
  const bool prepared = c->chain_prepared && c->gd_is_s;
  c->chain_prepared = false;
  PMFCHK(prc);
  const int64_t ldp = (int64_t)c->np + c->KP;
  if (!c->gd_is_s) {           // (the sums were cached, or crossed the ranks after the local reduce)
    pmf_launch_hessian_from_ps(c->stream, c->dPS, ldp, c->np, c->KP, c->k, c->dGd);
    HIPCHK(c, hipGetLastError());
  }
  // problems = columns: f[kk] = PS[kk][col] (stride ldp over kk, 1 over problems)
  PMFCHK(solve_nnqps(c, c->dPS, ldp, 1, c->dH, c->np, 1, c->n, false, prepared));
  c->g_valid = false; c->g_parts = 0; c->num_valid = false;
  c->ps_valid = true;
  c->trace_ready = false;
  return PMF_OK;
}

int do_update_w(pmf_ctx* c) {
  c->w_implicit = false;        // about to be overwritten (SNMF) -- only SNMF loops leave it set
  c->ps_valid = false;
  c->psd_fresh = false;
  c->trace_ready = false;
  switch (c->algo) {
    case PMF_ALGO_NMF: return nmf_update_w(c);
    case PMF_ALGO_BNMF: return nmf_update_w(c);
    case PMF_ALGO_RNMF: return nmf_update_w(c);
    case PMF_ALGO_SNMF: return snmf_update_w(c);
    case PMF_ALGO_NMFALS: return als_update_w(c);
  }
  return fail(c, PMF_EINVAL, "bad algo");
}

int do_update_h(pmf_ctx* c) {
  switch (c->algo) {
    case PMF_ALGO_NMF: return nmf_update_h(c);
    case PMF_ALGO_BNMF: return nmf_update_h(c);
    case PMF_ALGO_RNMF: return nmf_update_h(c);
    case PMF_ALGO_SNMF: return snmf_update_h(c);
    case PMF_ALGO_NMFALS: return als_update_h(c);
  }
  return fail(c, PMF_EINVAL, "bad algo");
}

int frobenius_direct(pmf_ctx* c, double* out) {
  PMFCHK(materialize_w(c));
  if (c->v_csr) return fail(c, PMF_EINVAL, "frobenius on CSR data: the reference returns its -123456 sentinel (nmf.py:109-112)");
  const int nb = (int)(c->mp / 64);
  PMFCHK(launch_resid(c, false, 0.f));
  hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(256), 0, c->stream, c->dPart, c->resid_parts, c->dScal);
  HIPCHK(c, hipGetLastError());
  PMFCHK(allreduce_sum(c, c->dScal, 1, true));
  double ss = 0.0;
  HIPCHK(c, hipMemcpyAsync(&ss, c->dScal, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *out = std::sqrt(ss);
  return PMF_OK;
}

// sum(V^2) over this rank's rows -> dScal[6]; enqueued right behind the upload of a dense V, so the
// first error evaluation does not pay a pass over V
int local_vnorm(pmf_ctx* c) {
  const int nb = 1024;
  hipLaunchKernelGGL(k_sumsq, dim3(nb), dim3(256), 0, c->stream, c->dV, (int64_t)c->mp * c->np, c->dPart);
  HIPCHK(c, hipGetLastError());
  hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(256), 0, c->stream, c->dPart, nb, c->dScal + 6);
  HIPCHK(c, hipGetLastError());
  c->vnorm_local_valid = true;
  return PMF_OK;
}

int ensure_vnorm(pmf_ctx* c) {
  if (c->vnorm_valid) return PMF_OK;
  if (!c->vnorm_local_valid) PMFCHK(local_vnorm(c));
  HIPCHK(c, hipMemcpyAsync(c->dScal, c->dScal + 6, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  PMFCHK(allreduce_sum(c, c->dScal, 1, true));
  HIPCHK(c, hipMemcpyAsync(&c->vnorm2, c->dScal, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->vnorm_valid = true;
  return PMF_OK;
}

// sqrt(sum((V - W H)^2)) (nmf.py:110).  When the partial sums P = W^T V, S = W^T W of the current
// W are at hand (every iteration that ran update_h), the trace identity
// ||V||^2 - 2<P,H> + <S H,H> gives the same number from k x n sized data in float64 -- no third
// pass over V and, across ranks, no extra collective (P, S are already all-reduced).  The identity
// cancels when the fit is nearly exact; below 1e-3 relative residual energy the direct pass runs.
// part[2 b], part[2 b + 1] = this column block's share of <P, H>, <S H, H> (k_trace_terms): from the float32 (P | S) and H, or --
// SNMF with its float64 H -- from Hd, and inside the Gram-space loop from the float64 P, S of the iteration at hand (psd_fresh)
int ensure_hd(pmf_ctx* c);
int launch_trace_terms(pmf_ctx* c) {
  const int nb = c->np / 16;
  const int64_t ldp = (int64_t)c->np + c->KP;
  if (c->algo == PMF_ALGO_SNMF && c->nb == 1 && c->opt_snmf_h64 != 0) {
    PMFCHK(ensure_hd(c));
    const size_t smem = (size_t)c->KP * 16 * sizeof(double);
    if (c->psd_fresh && c->ps_valid)
      hipLaunchKernelGGL((k_trace_terms<double, double>), dim3(nb), dim3(256), smem, c->stream, (const double*)c->dHd, (int64_t)c->np, c->np, c->KP,
                         (const double*)c->dPd, (int64_t)c->np, (const double*)c->dSd, (int64_t)c->KP, c->dPart);
    else
      hipLaunchKernelGGL((k_trace_terms<double, float>), dim3(nb), dim3(256), smem, c->stream, (const double*)c->dHd, (int64_t)c->np, c->np, c->KP,
                         (const float*)c->dPS, ldp, (const float*)c->dPS + c->np, ldp, c->dPart);
  } else {
    hipLaunchKernelGGL((k_trace_terms<float, float>), dim3(nb), dim3(256), (size_t)c->KP * 16 * sizeof(float), c->stream, (const float*)c->dH,
                       (int64_t)c->np, c->np, c->KP, (const float*)c->dPS, ldp, (const float*)c->dPS + c->np, ldp, c->dPart);
  }
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

int trace_e2(pmf_ctx* c, double* e2_out) {   // needs ps_valid and vnorm_valid
  double t[2] = {0.0, 0.0};
  if (c->trace_ready && c->ps_valid && c->trace_parts > 0) {   // ... as per-workgroup pairs
    double tp[2 * PMF_HGRAM_MAX_WGS];
    HIPCHK(c, hipMemcpyAsync(tp, c->dT1part, (size_t)2 * c->trace_parts * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int q = 0; q < c->trace_parts; ++q) { t[0] += tp[2 * q]; t[1] += tp[2 * q + 1]; }
  } else if (c->trace_ready && c->ps_valid) {   // the H-step kernel already produced both terms
    HIPCHK(c, hipMemcpyAsync(t, c->dScal + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  } else {
    const int nb = c->np / 16;
    PMFCHK(launch_trace_terms(c));
    hipLaunchKernelGGL(k_sum_pairs_f64, dim3(1), dim3(256), 0, c->stream, c->dPart, nb, c->dScal);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(t, c->dScal, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *e2_out = c->vnorm2 - 2.0 * t[0] + t[1];
  return PMF_OK;
}

int ensure_ps(pmf_ctx* c);

int do_frobenius(pmf_ctx* c, double* out) {
  if (c->nb > 1) {            // num_bases > 128: the residual through the trace identity (no MFMA residual kernel at that width)
    if (c->v_csr) return fail(c, PMF_EINVAL, "frobenius on CSR data: the reference returns its -123456 sentinel (nmf.py:109-112)");
    if (c->algo != PMF_ALGO_RNMF) {       // (RNMF's (P | S) are contractions with D = S - data, not with V)
      PMFCHK(ensure_ps(c));
      PMFCHK(ensure_vnorm(c));
      double e2 = 0.0;
      PMFCHK(trace_e2(c, &e2));
      if (e2 > 1e-3 * c->vnorm2) { *out = std::sqrt(e2); return PMF_OK; }
    }
    // the identity cancels: direct pass (plain FMAs; num_bases > 128 has no MFMA residual kernel)
    PMFCHK(materialize_w(c));
    PMFCHK(resid_bigk(c, false, c->dScal));
    PMFCHK(allreduce_sum(c, c->dScal, 1, true));
    double ss = 0.0;
    HIPCHK(c, hipMemcpyAsync(&ss, c->dScal, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out = std::sqrt(ss);
    return PMF_OK;
  }
  if (c->v_csr || !c->ps_valid) return frobenius_direct(c, out);
  PMFCHK(ensure_vnorm(c));
  double e2 = 0.0;
  PMFCHK(trace_e2(c, &e2));
  if (!(e2 > 1e-3 * c->vnorm2)) return frobenius_direct(c, out);
  *out = std::sqrt(e2);
  return PMF_OK;
}

// Staging area for the host <-> device transport (grown on demand, at most kStageBytes at a time)
constexpr size_t kStageBytes = (size_t)256 << 20;
int stage_reserve(pmf_ctx* c, size_t bytes) {
  if (c->stage_cap >= bytes) return PMF_OK;
  bytes = std::max<size_t>(bytes, (size_t)4 << 20);          // (H and other k x n sized arrays: one allocation serves them all)
  if (c->dStage) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->dStage)); c->dStage = nullptr; c->stage_cap = 0; }
  HIPCHK(c, hipMalloc(&c->dStage, bytes));
  c->stage_cap = bytes;
  return PMF_OK;
}

// Host [rows][cols] (leading dimension sld, float32 or float64) -> device [rows][dld] float32, zero padded.  Contiguous host
// rows go up as they are in ONE hipMemcpyAsync per chunk (56 GB/s from pageable memory; hipMemcpy2DAsync: 17) and are padded /
// rounded by k_unpack_rows on the device; only a host array with a leading dimension of its own takes the pitched copy.
template <typename T>
int upload_rows(pmf_ctx* c, float* dst, int64_t dld, const T* src, int64_t sld, int64_t rows, int64_t cols) {
  constexpr bool f32 = sizeof(T) == sizeof(float);
  if (rows <= 0 || cols <= 0) return PMF_OK;
  if (f32 && sld == cols && dld == cols) {                    // nothing to pad, nothing to round
    HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)rows * cols * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PMF_OK;
  }
  if (f32 && sld != cols) {                                   // a pitched host array
    HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)dld * sizeof(float), src, (size_t)sld * sizeof(float),
                               (size_t)cols * sizeof(float), (size_t)rows, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PMF_OK;
  }
  const int64_t chunk_rows = std::max<int64_t>(1, std::min<int64_t>(rows, (int64_t)(kStageBytes / ((size_t)cols * sizeof(T)))));
  PMFCHK(stage_reserve(c, (size_t)chunk_rows * cols * sizeof(T)));
  for (int64_t r0 = 0; r0 < rows; r0 += chunk_rows) {
    const int64_t nr = std::min(chunk_rows, rows - r0);
    if (sld == cols)
      HIPCHK(c, hipMemcpyAsync(c->dStage, src + r0 * sld, (size_t)nr * cols * sizeof(T), hipMemcpyHostToDevice, c->stream));
    else
      HIPCHK(c, hipMemcpy2DAsync(c->dStage, (size_t)cols * sizeof(T), src + r0 * sld, (size_t)sld * sizeof(T), (size_t)cols * sizeof(T),
                                 (size_t)nr, hipMemcpyHostToDevice, c->stream));
    const unsigned grid = (unsigned)std::min<int64_t>((nr * dld + 255) / 256, 8192);
    hipLaunchKernelGGL((k_unpack_rows<T>), dim3(grid), dim3(256), 0, c->stream, reinterpret_cast<const T*>(c->dStage), nr, cols, dst + r0 * dld, dld);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PMF_OK;
}

int upload_padded(pmf_ctx* c, float* dst, int64_t dld, const float* src, int64_t sld, int64_t rows, int64_t cols) {
  return upload_rows<float>(c, dst, dld, src, sld, rows, cols);
}

// device [rows][sld] float32 -> host [rows][cols] contiguous float32 / float64 (the rounding to the host array's float64 on
// the device: np.copyto(float64, float32) of a 1 048 576 x 64 W is 0.1 s in one host thread)
template <typename T>
int download_rows(pmf_ctx* c, T* dst, const float* src, int64_t sld, int64_t rows, int64_t cols) {
  constexpr bool f32 = sizeof(T) == sizeof(float);
  if (rows <= 0 || cols <= 0) return PMF_OK;
  if (f32 && sld == cols) {
    HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)rows * cols * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PMF_OK;
  }
  const int64_t chunk_rows = std::max<int64_t>(1, std::min<int64_t>(rows, (int64_t)(kStageBytes / ((size_t)cols * sizeof(T)))));
  PMFCHK(stage_reserve(c, (size_t)chunk_rows * cols * sizeof(T)));
  for (int64_t r0 = 0; r0 < rows; r0 += chunk_rows) {
    const int64_t nr = std::min(chunk_rows, rows - r0);
    const unsigned grid = (unsigned)std::min<int64_t>((nr * cols + 255) / 256, 8192);
    hipLaunchKernelGGL((k_pack_rows<T>), dim3(grid), dim3(256), 0, c->stream, src + r0 * sld, sld, nr, cols, reinterpret_cast<T*>(c->dStage));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(dst + r0 * cols, c->dStage, (size_t)nr * cols * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));             // the staging area is reused by the next chunk
  }
  return PMF_OK;
}

int download_padded(pmf_ctx* c, float* dst, int64_t dld, const float* src, int64_t sld, int64_t rows, int64_t cols) {
  if (dld == cols) return download_rows<float>(c, dst, src, sld, rows, cols);
  HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)dld * sizeof(float), src, (size_t)sld * sizeof(float),
                             (size_t)cols * sizeof(float), (size_t)rows, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PMF_OK;
}

// Which launch site pmf_profile_enable times, with the ALGORITHMIC flops / bytes of ONE launch on THIS
// rank's rows and the flops the kernel really executes (symmetry of W^T W,
// reassociations) next to them.
void choose_stat_site(pmf_ctx* c, bool gram) {
  const double m = (double)c->m, n = (double)c->n, k = (double)c->k, nnz = (double)c->nnz;
  KernelStat& st = c->stat;
  const int old_site = st.site;
  st.site = SITE_NONE; st.name = "none"; st.flops = st.bytes = st.exec_flops = 0.0;
  char buf[96];
  if (c->algo == PMF_ALGO_SNMF && gram) {
    st.site = SITE_MATERIALIZE;                   // the only m-sized kernel of a Gram-space loop: W = V M, once
    if (use_csr(c)) {
      st.name = "k_csr_w_blocks(W = V M)";
      st.flops = st.exec_flops = 2.0 * nnz * k;
      st.bytes = 4.0 * m * k + 8.0 * nnz + 8.0 * (m + 1.0);      // W written once; CSR arrays read once
    } else {
      snprintf(buf, sizeof(buf), "k_rowgemm<%d,store>(W = V M^T)", c->NT);
      st.name = buf;
      st.flops = st.exec_flops = 2.0 * m * n * k;
      st.bytes = 4.0 * (m * n + m * k);
    }
  } else if (c->algo == PMF_ALGO_SNMF && use_csr(c)) {
    st.site = SITE_CSR_PASS;
    st.name = "k_snmf_csr_mfma (one pass per iteration)";
    st.flops = 4.0 * nnz * k + 4.0 * m * k * k;                  // SURVEY: SpMM, (.) inv, W^T V, W^T W
    st.exec_flops = 4.0 * nnz * k + m * k * (k + 16.0);          // V M, W^T V, upper triangle of W^T W
    st.bytes = 4.0 * m * k + 8.0 * nnz + 8.0 * (m + 1.0);
  } else if (c->fused_wgs > 0 && c->algo != PMF_ALGO_NMFALS) {
    st.site = SITE_FUSED;
    st.name = c->fused8 ? c->path.c_str()
                        : pmf_fused_kernel_name(c->NT, c->np, c->algo == PMF_ALGO_SNMF   ? FUSED_SNMF
                                                          : c->algo == PMF_ALGO_BNMF ? FUSED_BNMF
                                                          : c->algo == PMF_ALGO_RNMF ? FUSED_RNMF
                                                                                     : FUSED_NMF);
    // one pass over V does the four m-sized contractions of an iteration: F = 4 m n k + 4 m k^2
    st.flops = 4.0 * m * n * k + 4.0 * m * k * k;
    if (c->algo == PMF_ALGO_SNMF) {               // executes V M^T, W^T V and the upper triangle of W^T W
      st.exec_flops = 4.0 * m * n * k + m * k * (k + 16.0);
      st.bytes = 4.0 * (m * n + m * k);           // V read once, W written once
    } else if (c->fused8) {                       // V H^T, W G, W^T V and ALL of W^T W (base split: no symmetry to use)
      st.exec_flops = 4.0 * m * n * k + 4.0 * m * k * k;
      st.bytes = 4.0 * (m * n + 2.0 * m * k);
    } else {                                      // V H^T, W G, W^T V and the upper triangle of W^T W
      st.exec_flops = 4.0 * m * n * k + 2.0 * m * k * k + m * k * (k + 16.0);
      st.bytes = 4.0 * (m * n + 2.0 * m * k);     // V read once, W read and written once
    }
  } else if (c->algo == PMF_ALGO_NMFALS) {
    st.site = SITE_NNQP_W;
    if (c->opt_nnqp_quad && c->k <= 64 && (c->m >= 16384 || c->opt_nnqp_quad == 2)) snprintf(buf, sizeof(buf), "k_nnqp_quad(update_w)");
    else if (c->k <= 64) snprintf(buf, sizeof(buf), "k_nnqp<%d>(update_w)", c->k <= 16 ? 16 : c->k <= 32 ? 32 : 64);
    else if (nnqp_use_wave(c)) snprintf(buf, sizeof(buf), "k_nnqp_wave(update_w)");
    else snprintf(buf, sizeof(buf), "k_nnqp_big<%d>(update_w)", pmf_nnqp_big_vpl(c->k));
    st.name = buf;
    st.bytes = 4.0 * (3.0 * m * k);               // right-hand sides read, warm start read, solution written
  } else if ((c->algo == PMF_ALGO_NMF) && c->nb == 1) {
    st.site = SITE_ROWGEMM_W;
    snprintf(buf, sizeof(buf), "k_rowgemm<%d,nmf_w>", c->NT);
    st.name = buf;
    st.flops = st.exec_flops = 2.0 * m * n * k + 2.0 * m * k * k;
    st.bytes = 4.0 * (m * n + 2.0 * m * k);
  }
  if (st.site != old_site) st.used = 0;
}

}  // namespace

// =============================================================================================
extern "C" {

int pmf_device_count(int32_t* out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { g_create_error = hipGetErrorString(e); *out = 0; return PMF_EHIP; }
  *out = n;
  return PMF_OK;
}

int pmf_nccl_unique_id(void* out) {
  static_assert(sizeof(ncclUniqueId) == PMF_NCCL_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) { g_create_error = ncclGetErrorString(r); return PMF_ENCCL; }
  std::memcpy(out, &id, sizeof(id));
  return PMF_OK;
}

int pmf_ctx_create(pmf_ctx** out, int32_t algo, int64_t m_local, int64_t n, int32_t k, int32_t device,
                   int32_t rank, int32_t nranks, const void* nccl_id) {
  if (!out) return fail(nullptr, PMF_EINVAL, "out is NULL");
  *out = nullptr;
  if (algo < 0 || algo > 4) return fail(nullptr, PMF_EINVAL, "algo must be 0 (NMF), 1 (NMFALS), 2 (SNMF), 3 (BNMF) or 4 (RNMF)");
  if (m_local < 1 || n < 1 || k < 1) return fail(nullptr, PMF_EINVAL, "m, n, k must be >= 1");
  // The reference has no limit on num_bases (nmf.py:116-120); the generic kernels beyond 128 bases have been checked against
  // the float64 oracles at 1 500, 2 304 and 2 432 bases (tests/sweeps/bigk_limit_probe.py, tests/test_gpu_bigk.py); beyond 2 432 (19 blocks of 128)
  // the 16-column block of H that k_nmf_h and k_trace_terms keep in LDS (64 bytes per basis) no longer fits the 160 KiB.  NMFALS /
  // NMFNNLS beyond 128 bases solve one variable at a time (k_nnqp_big) and stay at 1 024.
  if (k > (algo == PMF_ALGO_NMFALS ? 1024 : 2432))
    return fail(nullptr, PMF_EINVAL, algo == PMF_ALGO_NMFALS ? "num_bases > 1024 is not supported for NMFALS / NMFNNLS by this build"
                                                             : "num_bases > 2432 is not supported by this build");
  if (n > (1 << 24)) return fail(nullptr, PMF_EINVAL, "n too large");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(nullptr, PMF_EINVAL, "bad rank/nranks");
  if (nranks > 1 && !nccl_id) return fail(nullptr, PMF_EINVAL, "nccl_id required when nranks > 1");
  pmf_ctx* c = new (std::nothrow) pmf_ctx();
  if (!c) return fail(nullptr, PMF_ENOMEM, "host allocation failed");
  c->algo = algo; c->m = m_local; c->n = n; c->k = k; c->device = device; c->rank = rank; c->nranks = nranks;
  c->mp = round_up(m_local, 64);
  c->np = (int)round_up(n, 64);
  c->NT = k <= 16 ? 1 : k <= 32 ? 2 : k <= 64 ? 4 : 8;
  c->KP = 16 * c->NT;
  if (k > 128) {                // blocks of 128 bases on the NT = 8 tiled kernels (bigk_* below)
    c->nb = (int)((k + 127) / 128);
    c->KP = 128 * c->nb;
  }
  // 448 columns are not a panel count the wide (two waves per block) fused kernel takes: pad to 512
  if ((algo == PMF_ALGO_NMF || algo == PMF_ALGO_BNMF) && c->NT <= 2 && c->np == 448) c->np = 512;
  // ... and the cooperative kernel (pmf_coop.h) takes 6, 8, 12 or 16 column panels beyond 4
  if ((algo == PMF_ALGO_NMF || algo == PMF_ALGO_BNMF) && c->nb == 1 && k <= 128) {
    const int padded = pmf_coop_pad_np(c->NT, c->np);
    if (padded > 0) c->np = padded;
  }
  int rc = [&]() -> int {
    HIPCHK(c, hipSetDevice(device));
    HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(c, hipEventCreate(&c->ev0));
    HIPCHK(c, hipEventCreate(&c->ev1));
    if (nccl_id) {   // nranks == 1 with an id: a 1-rank communicator (exercises the RCCL path)
      ncclUniqueId id;
      std::memcpy(&id, nccl_id, sizeof(id));
      NCCLCHK(c, ncclCommInitRank(&c->comm, nranks, id, rank));
    }
    // partial-slab geometry for the tiled W^T V path: ~1024 row chunks over the grid
    const int n_panels = (c->np + 255) / 256;
    int want = std::max(1, 1024 / n_panels);
    int64_t blocks16 = c->mp / 16;
    c->nchunks = (int)std::min<int64_t>(want, blocks16);
    c->rows_per_chunk = (int)(round_up((blocks16 + c->nchunks - 1) / c->nchunks, 4) * 16);   // whole 64-row stages (k_colgemm_stream)
    c->nchunks = (int)((c->mp + c->rows_per_chunk - 1) / c->rows_per_chunk);
    c->fused_wgs = (c->nb == 1 && (algo == PMF_ALGO_NMF || algo == PMF_ALGO_SNMF || algo == PMF_ALGO_BNMF || algo == PMF_ALGO_RNMF))
                       ? pmf_fused_grid_for(c->NT, c->np, c->mp, /*allow_split=*/algo != PMF_ALGO_SNMF) : 0;
    {
      int bt = 0, rb = 0, pn = 0;
      if (c->fused_wgs == 0 && c->nb == 1 && pmf_coop_shape(c->NT, c->np, &bt, &rb, &pn) &&
          (algo == PMF_ALGO_NMF || algo == PMF_ALGO_BNMF || (algo == PMF_ALGO_RNMF && bt == 2 && rb == 4))) {
        c->fused8 = true;
        c->coop_bt = bt; c->coop_rb = rb;
        c->fused_wgs = pmf_coop_grid_for(c->mp, rb);
      }
    }
    const int nslabs = std::max(c->nchunks, c->fused_wgs);
    // dV [mp][np] is allocated by the first pmf_set_v_dense_f32 / pmf_fill_v_uniform: CSR and
    // streamed (pmf_stream_*) contexts never hold a dense V
    PMFCHK(dalloc(c, &c->dW, (size_t)c->mp * c->KP));
    PMFCHK(dalloc(c, &c->dH, (size_t)c->KP * c->np));
    PMFCHK(dalloc(c, &c->dG, (size_t)c->KP * c->KP));
    PMFCHK(dalloc(c, &c->dGd, (size_t)c->KP * c->KP));
    PMFCHK(dalloc(c, &c->dPS, (size_t)ps_elems(c)));
    if (c->nb == 1) {
      PMFCHK(dalloc(c, &c->dSlab, (size_t)nslabs * ps_elems(c)));
    } else {                    // one 128-base block at a time: [chunk][128][max(np, KP) + 128]
      PMFCHK(dalloc(c, &c->dSlab, (size_t)c->nchunks * 128 * (std::max(c->np, c->KP) + 128)));
      PMFCHK(dalloc(c, &c->dW1, (size_t)std::max<int64_t>(c->mp, c->np) * c->KP));
      PMFCHK(dalloc(c, &c->dW2, (size_t)c->mp * c->KP));
    }
    c->dpart_cap = std::max<int64_t>(std::max<int64_t>(c->mp / 64, 1024), c->np / 8 + 2);
    PMFCHK(dalloc(c, &c->dPart, (size_t)c->dpart_cap));
    PMFCHK(dalloc(c, &c->dScal, 8));
    PMFCHK(dalloc(c, &c->dGpart, (size_t)PMF_HGRAM_MAX_WGS * c->KP * c->KP));
    PMFCHK(dalloc(c, &c->dT1part, (size_t)2 * PMF_HGRAM_MAX_WGS));
    PMFCHK(dalloc(c, &c->dTicket, 1));
    PMFCHK(dalloc(c, &c->dStop, 2));
    c->ferr_cap = 4096;
    PMFCHK(dalloc(c, &c->dFerr, (size_t)c->ferr_cap));
    if (algo == PMF_ALGO_RNMF) PMFCHK(dalloc(c, &c->dD, (size_t)c->mp * c->np));
    if (algo != PMF_ALGO_NMF) {
      if (!c->dW1) PMFCHK(dalloc(c, &c->dW1, (size_t)std::max<int64_t>(c->mp, c->np) * c->KP));
      PMFCHK(dalloc(c, &c->dGinvT, (size_t)c->KP * c->KP));
    }
    if (algo == PMF_ALGO_SNMF) {
      PMFCHK(dalloc(c, &c->dMT, (size_t)c->KP * c->np));
      PMFCHK(dalloc(c, &c->dGinvD, (size_t)c->KP * c->KP));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PMF_OK;
  }();
  if (rc != PMF_OK) {
    g_create_error = c->err;
    pmf_ctx_destroy(c);
    return rc;
  }
  if (c->fused8) {
    char nb_[64];
    snprintf(nb_, sizeof(nb_), "k_nmf_coop<%d,%d,%d%s>", c->coop_bt, c->coop_rb, c->np / 64,
             algo == PMF_ALGO_BNMF ? ",bnmf" : algo == PMF_ALGO_RNMF ? ",rnmf" : "");
    c->path = nb_;
  } else
  c->path = (c->fused_wgs > 0) ? std::string(pmf_fused_kernel_name(c->NT, c->np, algo == PMF_ALGO_SNMF   ? FUSED_SNMF
                                                                           : algo == PMF_ALGO_BNMF ? FUSED_BNMF
                                                                           : algo == PMF_ALGO_RNMF ? FUSED_RNMF
                                                                                                   : FUSED_NMF))
                               : std::string("tiled");
  choose_stat_site(c, false);
  *out = c;
  return PMF_OK;
}

int pmf_ctx_destroy(pmf_ctx* c) {
  if (!c) return PMF_OK;
  (void)hipSetDevice(c->device);   // teardown: there is nobody to report a failing release to
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  if (c->ipc_exported) {
    for (int r = 0; r < PMF_IPC_MAX_RANKS; ++r)
      if (r != c->ipc.me && c->ipc.area[r]) (void)hipIpcCloseMemHandle(c->ipc.area[r]);
    if (c->ipc.area[c->ipc.me]) (void)hipFree(c->ipc.area[c->ipc.me]);
  }
  if (c->dIpcErr) (void)hipFree(c->dIpcErr);
  if (c->dHsnap) (void)hipFree(c->dHsnap);
  if (c->dIpcWait) (void)hipFree(c->dIpcWait);
  if (c->dIpcTestA) (void)hipFree(c->dIpcTestA);
  if (c->dIpcTestB) (void)hipFree(c->dIpcTestB);
  for (void* p : {(void*)c->dV, (void*)c->dW, (void*)c->dH, (void*)c->dG, (void*)c->dPS, (void*)c->dSlab,
                  (void*)c->dW1, (void*)c->dGinvT, (void*)c->dD, (void*)c->dGd, (void*)c->dPart, (void*)c->dScal,
                  (void*)c->dIndptr, (void*)c->dIndices, (void*)c->dVals})
    if (p) (void)hipFree(p);
  for (void* p : {(void*)c->dTile[0], (void*)c->dTile[1], (void*)c->dPSacc, (void*)c->dStAcc, (void*)c->dGpart,
                  (void*)c->dT1part, (void*)c->dTicket, (void*)c->dFerr, (void*)c->dStop, (void*)c->dWarm, (void*)c->dW2,
                  (void*)c->dMT, (void*)c->dGinvD, (void*)c->dC, (void*)c->dCslabs, (void*)c->dMTd, (void*)c->dPd, (void*)c->dInvA, (void*)c->dInvB, (void*)c->dQp, (void*)c->dSing, (void*)c->dBinv, (void*)c->dDefer, (void*)c->dNbig, (void*)c->dWsnap, (void*)c->dY0, (void*)c->dQstat, c->dStage, (void*)c->dGramPart, (void*)c->dGramTickets, (void*)c->dWideT, (void*)c->dWideN, (void*)c->dWideD, (void*)c->dHd, (void*)c->dSd, (void*)c->dHdSnap, (void*)c->dVmaxBits})
    if (p) (void)hipFree(p);
  for (hipEvent_t e : {c->ev_copied[0], c->ev_copied[1], c->ev_consumed[0], c->ev_consumed[1]})
    if (e) (void)hipEventDestroy(e);
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->w_stream) { (void)hipStreamSynchronize(c->w_stream); (void)hipStreamDestroy(c->w_stream); }
  for (hipEvent_t e : {c->ev_mt[0], c->ev_mt[1], c->ev_w[0], c->ev_w[1]}) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->stat.ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->coll_ev) (void)hipEventDestroy(e);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return PMF_OK;
}

const char* pmf_last_error(const pmf_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
const char* pmf_path_name(const pmf_ctx* c) { return c ? c->path.c_str() : ""; }

// RNMF keeps D = S - data on the device (rnmf.py:102,111); the reference's S is an attribute that SURVIVES new data
// (update_w / update_h then work on S - new data until the next update_s): around a change of V the state goes
// D -> S = D + V_old -> D = S - V_new.
static int rnmf_data_change_begin(pmf_ctx* c) {
  if (c->algo != PMF_ALGO_RNMF || !c->s_valid || !c->have_v || !c->dD || !c->dV) return PMF_OK;
  const int64_t E = c->mp * c->np;
  hipLaunchKernelGGL(k_acc_f32, dim3(elem_grid(E / 4)), dim3(256), 0, c->stream, c->dD, c->dV, E);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}
static int rnmf_data_change_end(pmf_ctx* c) {
  if (c->algo != PMF_ALGO_RNMF || !c->s_valid || !c->dD || !c->dV) return PMF_OK;
  const int64_t E = c->mp * c->np;
  hipLaunchKernelGGL(k_sub_f32, dim3(elem_grid(E / 4)), dim3(256), 0, c->stream, c->dD, c->dV, E);
  HIPCHK(c, hipGetLastError());
  return PMF_OK;
}

int pmf_set_v_dense_f32(pmf_ctx* c, const float* V, int64_t ld) {
  if (!c || !V || ld < c->n) return fail(c, PMF_EINVAL, "pmf_set_v_dense_f32: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  PMFCHK(ensure_dv(c));
  PMFCHK(rnmf_data_change_begin(c));
  PMFCHK(upload_padded(c, c->dV, c->np, V, ld, c->m, c->n));
  PMFCHK(rnmf_data_change_end(c));
  c->have_v = true; c->v_csr = false; c->csr_dense = false; c->vnorm_valid = false; c->ps_valid = false; c->num_valid = false; c->c_valid = false;
  PMFCHK(local_vnorm(c));
  return PMF_OK;
}

int pmf_set_v_csr_f32(pmf_ctx* c, const int64_t* indptr, const int32_t* indices, const float* vals,
                      int64_t nnz) {
  if (!c || !indptr || (nnz > 0 && (!indices || !vals)) || nnz < 0)
    return fail(c, PMF_EINVAL, "pmf_set_v_csr_f32: bad arguments");
  if (c->algo != PMF_ALGO_SNMF) return fail(c, PMF_EINVAL, "CSR input is only wired for SNMF");
  HIPCHK(c, hipSetDevice(c->device));
  for (void* p : {(void*)c->dIndptr, (void*)c->dIndices, (void*)c->dVals}) if (p) HIPCHK(c, hipFree(p));
  c->dIndptr = nullptr; c->dIndices = nullptr; c->dVals = nullptr;
  PMFCHK(dalloc(c, &c->dIndptr, (size_t)c->mp + 1));
  PMFCHK(dalloc(c, &c->dIndices, (size_t)nnz));
  PMFCHK(dalloc(c, &c->dVals, (size_t)nnz));
  std::vector<int64_t> ip((size_t)c->mp + 1);
  for (int64_t r = 0; r <= c->mp; ++r) ip[(size_t)r] = indptr[std::min(r, c->m)];
  HIPCHK(c, hipMemcpyAsync(c->dIndptr, ip.data(), ip.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  if (nnz) {
    HIPCHK(c, hipMemcpyAsync(c->dIndices, indices, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dVals, vals, (size_t)nnz * sizeof(float), hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->nnz = nnz; c->have_v = true; c->v_csr = true; c->vnorm_valid = false; c->vnorm_local_valid = false; c->ps_valid = false; c->num_valid = false;
  c->c_valid = false;            // V^T V for the Gram-space loop is formed on first use (k_csr_gram)
  c->csr_dense = false;
  // num_bases > 128: no CSR kernel at that width; n * num_bases beyond the 160 KiB LDS accumulator of the CSR scatter
  // (k_csr_p: e.g. 520 columns x 100 bases): no CSR kernel at that SIZE -- the rows are expanded once
  if (c->nb > 1 || (size_t)c->np * c->KP * sizeof(float) > (size_t)160 * 1024) {
    PMFCHK(ensure_dv(c));
    HIPCHK(c, hipMemsetAsync(c->dV, 0, (size_t)c->mp * c->np * sizeof(float), c->stream));
    hipLaunchKernelGGL(k_csr_densify, dim3((unsigned)((c->m + 255) / 256)), dim3(256), 0, c->stream, c->dIndptr, c->dIndices,
                       c->dVals, c->m, c->np, c->dV);
    HIPCHK(c, hipGetLastError());
    c->csr_dense = true;
    PMFCHK(local_vnorm(c));
  }
  return PMF_OK;
}

static int fill(pmf_ctx* c, float* X, int64_t ld, int64_t rows, int64_t cols, int64_t row0, uint64_t seed) {
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t E = rows * cols;
  hipLaunchKernelGGL(k_fill_uniform, dim3(elem_grid(E)), dim3(256), 0, c->stream, X, ld,
                     rows, cols, row0, cols, seed);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PMF_OK;
}

int pmf_fill_v_uniform(pmf_ctx* c, uint64_t seed, int64_t row0) {
  if (!c) return PMF_EINVAL;
  PMFCHK(ensure_dv(c));
  PMFCHK(rnmf_data_change_begin(c));
  PMFCHK(fill(c, c->dV, c->np, c->m, c->n, row0, seed));
  PMFCHK(rnmf_data_change_end(c));
  c->have_v = true; c->v_csr = false; c->csr_dense = false; c->vnorm_valid = false; c->ps_valid = false; c->num_valid = false; c->c_valid = false;
  PMFCHK(local_vnorm(c));
  return PMF_OK;
}
int pmf_fill_w_uniform(pmf_ctx* c, uint64_t seed, int64_t row0) {
  if (!c) return PMF_EINVAL;
  PMFCHK(w_pipe_join(c));        // (a pipelined W = V M write still in flight on the side stream must not land on the new W)
  PMFCHK(fill(c, c->dW, c->KP, c->m, c->k, row0, seed));
  c->have_w = true; c->ps_valid = false; c->trace_ready = false;   // (the trace terms <P,H>, <S,G> belong to the old W as well)
  return PMF_OK;
}
int pmf_fill_h_uniform(pmf_ctx* c, uint64_t seed) {
  if (!c) return PMF_EINVAL;
  PMFCHK(fill(c, c->dH, c->np, c->k, c->n, 0, seed));
  c->have_h = true; c->g_valid = false; c->g_parts = 0; c->num_valid = false; c->trace_ready = false; c->hd_synced = false; c->hd_force = true;
  return PMF_OK;
}

// The padding of W ([mp][KP]) and H ([KP][np]) is zero by construction and stays zero under every update
// rule -- unless a factor went non-finite (0 * nan = nan), e.g. behind a singular H H^T.  A fresh factor
// from the host therefore comes with fresh padding.
static int zero_padding(pmf_ctx* c, float* buf, int64_t ld, int64_t rows_total, int64_t rows_valid, int64_t cols_valid) {
  if (ld > cols_valid && rows_valid > 0)
    HIPCHK(c, hipMemset2DAsync(buf + cols_valid, (size_t)ld * sizeof(float), 0, (size_t)(ld - cols_valid) * sizeof(float),
                               (size_t)rows_valid, c->stream));
  if (rows_total > rows_valid)
    HIPCHK(c, hipMemsetAsync(buf + rows_valid * ld, 0, (size_t)(rows_total - rows_valid) * ld * sizeof(float), c->stream));
  return PMF_OK;
}

int pmf_set_w_f32(pmf_ctx* c, const float* W) {
  if (!c || !W) return fail(c, PMF_EINVAL, "pmf_set_w_f32: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  PMFCHK(w_pipe_join(c));
  PMFCHK(zero_padding(c, c->dW, c->KP, c->mp, c->m, c->k));
  PMFCHK(upload_padded(c, c->dW, c->KP, W, c->k, c->m, c->k));
  c->have_w = true; c->ps_valid = false; c->trace_ready = false; c->w_implicit = false;
  return PMF_OK;
}
int pmf_set_w_f64(pmf_ctx* c, const double* W) {
  if (!c || !W) return fail(c, PMF_EINVAL, "pmf_set_w_f64: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  PMFCHK(w_pipe_join(c));
  PMFCHK(zero_padding(c, c->dW, c->KP, c->mp, c->m, c->k));
  PMFCHK(upload_rows<double>(c, c->dW, c->KP, W, c->k, c->m, c->k));
  c->have_w = true; c->ps_valid = false; c->trace_ready = false; c->w_implicit = false;
  return PMF_OK;
}
int pmf_get_w_f64(pmf_ctx* c, double* W) {
  PMFCHK(need(c, false, true, false));
  if (!W) return fail(c, PMF_EINVAL, "W is NULL");
  PMFCHK(materialize_w(c));
  return download_rows<double>(c, W, c->dW, c->KP, c->m, c->k);
}
int pmf_set_h_f64(pmf_ctx* c, const double* H) {
  if (!c || !H) return fail(c, PMF_EINVAL, "pmf_set_h_f64: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  PMFCHK(zero_padding(c, c->dH, c->np, c->KP, c->k, c->n));
  PMFCHK(upload_rows<double>(c, c->dH, c->np, H, c->n, c->k, c->n));
  c->have_h = true; c->g_valid = false; c->g_parts = 0; c->num_valid = false; c->trace_ready = false; c->hd_synced = false; c->hd_force = true;
  if (snmf_h64(c)) {               // SNMF: the caller's float64 H as it is (nmf.py:120 keeps H in float64), beside its rounding
    if (!c->dHd) { PMFCHK(dalloc(c, &c->dHd, (size_t)c->KP * c->np)); PMFCHK(dalloc(c, &c->dSd, (size_t)c->KP * c->KP)); }
    const size_t bytes = (size_t)c->k * c->n * sizeof(double);
    PMFCHK(stage_reserve(c, bytes));
    HIPCHK(c, hipMemsetAsync(c->dHd, 0, (size_t)c->KP * c->np * sizeof(double), c->stream));
    HIPCHK(c, hipMemcpyAsync(c->dStage, H, bytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_unpack_rows_f64, dim3((unsigned)std::min<int64_t>(((int64_t)c->k * c->np + 255) / 256, 1024)), dim3(256), 0, c->stream,
                       reinterpret_cast<const double*>(c->dStage), (int64_t)c->k, (int64_t)c->n, c->dHd, (int64_t)c->np);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->hd_synced = false; c->hd_force = false;   // (Hd holds the caller's float64 values, dH their rounding)
  }
  return PMF_OK;
}
int pmf_get_h_f64(pmf_ctx* c, double* H) {
  PMFCHK(need(c, false, false, true));
  if (!H) return fail(c, PMF_EINVAL, "H is NULL");
  if (snmf_h64(c) && c->dHd) {     // SNMF: the float64 H the device iterates on (entries another writer of the float32 H replaced: widened)
    PMFCHK(ensure_hd(c));
    const size_t bytes = (size_t)c->k * c->n * sizeof(double);
    PMFCHK(stage_reserve(c, bytes));
    hipLaunchKernelGGL(k_pack_rows_f64, dim3((unsigned)std::min<int64_t>(((int64_t)c->k * c->n + 255) / 256, 1024)), dim3(256), 0, c->stream,
                       (const double*)c->dHd, (int64_t)c->np, (int64_t)c->k, (int64_t)c->n, reinterpret_cast<double*>(c->dStage));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(H, c->dStage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return PMF_OK;
  }
  return download_rows<double>(c, H, c->dH, c->np, c->k, c->n);
}
int pmf_set_v_dense_f64(pmf_ctx* c, const double* V, int64_t ld) {
  if (!c || !V || ld < c->n) return fail(c, PMF_EINVAL, "pmf_set_v_dense_f64: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  PMFCHK(ensure_dv(c));
  PMFCHK(rnmf_data_change_begin(c));
  PMFCHK(upload_rows<double>(c, c->dV, c->np, V, ld, c->m, c->n));
  PMFCHK(rnmf_data_change_end(c));
  c->have_v = true; c->v_csr = false; c->csr_dense = false; c->vnorm_valid = false; c->ps_valid = false; c->num_valid = false; c->c_valid = false;
  PMFCHK(local_vnorm(c));
  return PMF_OK;
}

int pmf_get_w_f32(pmf_ctx* c, float* W) {
  PMFCHK(need(c, false, true, false));
  if (!W) return fail(c, PMF_EINVAL, "W is NULL");
  PMFCHK(materialize_w(c));
  return download_padded(c, W, c->k, c->dW, c->KP, c->m, c->k);
}
int pmf_set_h_f32(pmf_ctx* c, const float* H) {
  if (!c || !H) return fail(c, PMF_EINVAL, "pmf_set_h_f32: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  PMFCHK(zero_padding(c, c->dH, c->np, c->KP, c->k, c->n));
  PMFCHK(upload_padded(c, c->dH, c->np, H, c->n, c->k, c->n));
  c->have_h = true; c->g_valid = false; c->g_parts = 0; c->num_valid = false; c->trace_ready = false; c->hd_synced = false; c->hd_force = true;   // (P | S) do not depend on H
  return PMF_OK;
}
int pmf_get_h_f32(pmf_ctx* c, float* H) {
  PMFCHK(need(c, false, false, true));
  if (!H) return fail(c, PMF_EINVAL, "H is NULL");
  return download_padded(c, H, c->n, c->dH, c->np, c->k, c->n);
}

int pmf_update_w(pmf_ctx* c) {
  PMFCHK(need(c, true, true, true));
  PMFCHK(do_update_w(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  PMFCHK(ipc_check(c));
  return check_singular(c);
}
int pmf_update_h(pmf_ctx* c) {
  PMFCHK(need(c, true, true, true));
  PMFCHK(do_update_h(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return ipc_check(c);
}
int pmf_frobenius(pmf_ctx* c, double* out) {
  PMFCHK(need(c, true, true, true));
  if (!out) return fail(c, PMF_EINVAL, "out is NULL");
  PMFCHK(do_frobenius(c, out));
  return ipc_check(c);
}

int pmf_factorize(pmf_ctx* c, int32_t niter, uint32_t flags, double conv_eps, double* ferr,
                  int32_t* iters_done, int32_t* converged_at) {
  PMFCHK(need(c, true, true, true));
  const bool cw = flags & PMF_COMPUTE_W, ch = flags & PMF_COMPUTE_H, ce = flags & PMF_COMPUTE_ERR;
  if (niter < 0 || (ce && !ferr)) return fail(c, PMF_EINVAL, "pmf_factorize: bad arguments");
  // every early (error) return below leaves no pipelined W = V M write in flight on the side stream: a caller that then
  // re-uploads W must not see the stale product land on top of it
  struct WPipeGuard {
    pmf_ctx* c; bool ok = false;
    ~WPipeGuard() {
      if (ok || !c->w_stream) return;
      (void)hipStreamSynchronize(c->w_stream);
      c->ev_w_pending[0] = c->ev_w_pending[1] = false;
    }
  } wguard{c};
  if (iters_done) *iters_done = 0;
  if (converged_at) *converged_at = -1;
  c->want_trace = ce;
  c->fixed_h_loop = cw && !ch && niter > 1 && c->algo == PMF_ALGO_NMF;
  const bool fused = cw && ch && c->fused_wgs > 0 && !use_csr(c) &&
                     (c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_SNMF || c->algo == PMF_ALGO_BNMF ||
                      c->algo == PMF_ALGO_RNMF);
  HIPCHK(c, hipEventRecord(c->ev0, c->stream));
  int done = 0;
  // Free-running form of the loop (NMF on the fused kernel with the error on): after one
  // iteration in the ordinary form, chunks of iterations are enqueued back to back; the error and
  // the convergence test of nmf.py:134-139 run on the device (k_conv_check) and a raised stop flag
  // turns every later launch of the chunk into a no-op, so the results are those of the ordinary
  // loop while the host reads back once per chunk instead of once per iteration.
  // ... and the fixed-basis loop (compute_w = False, nmf.py:56-65: coefficients for an existing basis):
  // (W^T V | W^T W) is formed once, every further iteration is the H-step kernel alone
  const bool h_only = !cw && ch && ce && c->nb == 1 && !use_csr(c) &&
                      (c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_BNMF);
  // (a host transport for the cross-rank sums blocks on the host every iteration: nothing to free-run)
  // SNMF with both updates on: the loop runs in Gram space (snmf_gram_iteration), W materialised at the end
  const bool gram = cw && ch && snmf_gram_ok(c, niter);
  if (gram) PMFCHK(ensure_vgram(c));
  choose_stat_site(c, gram);
  const bool can_free_run = ((((fused && c->algo != PMF_ALGO_RNMF) || (gram && !use_csr(c) && c->nb == 1)) && ce) || h_only) &&
                            !(c->host_ar && !(c->ipc.nranks > 1 && (size_t)ps_elems(c) * sizeof(float) <= PMF_IPC_MAX_BYTES));   // NMF, BNMF, SNMF on the fused kernel
  // (a host transport blocks on the host in every iteration -- nothing to free-run -- unless the per-iteration payload
  //  (P | S) fits the one-shot IPC all-reduce in front of it)
  constexpr int kHostIters = 1, kChunk = 32;
  bool free_run = false;
  for (int i = 0; i < niter; ++i) {                       // nmf.py:182
    if (c->abort_flag.load(std::memory_order_relaxed) != 0) break;   // pmf_abort: the caller discards this run (iters_done says how far it got)
    if (free_run) {
      const int chunk = std::min(kChunk, niter - i);
      c->stop_arg = c->dStop;
      const double lamb_w0 = c->lamb_w, lamb_h0 = c->lamb_h;   // BNMF: every H step scales them (bnmf.py:84-85)
      const unsigned ipc_seq0 = c->ipc_seq;
      const long long fold_calls0 = c->fold_calls;
      unsigned seq_after[kChunk];                           // the exchange counter behind iteration i + j
      int lrc = PMF_OK;
      for (int j = 0; j < chunk && lrc == PMF_OK; ++j) {
        seq_after[j] = c->ipc_seq;                          // (overwritten below once the iteration is enqueued)
        c->gram_partial_ok = (c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_BNMF) && i + j + 1 < niter && !c->fused8;
        if (h_only) {
          lrc = ensure_ps(c);                               // current since the first iteration (W is fixed)
          if (lrc == PMF_OK) lrc = h_step_from_ps(c);
        } else {
          lrc = gram ? snmf_gram_iteration(c) : c->algo == PMF_ALGO_SNMF ? snmf_fused_iteration(c) : nmf_fused_iteration(c);
        }
        const double* tt = c->dScal + 2;                  // k_nmf_h_gram left <P,H>, <S,G> there ...
        int ntt = 1;
        if (lrc == PMF_OK && c->trace_ready && c->trace_parts > 0) { tt = c->dT1part; ntt = c->trace_parts; }   // ... or as pairs
        if (lrc == PMF_OK && !c->trace_ready) {           // SNMF: the H-step kernel does not form them
          const int nb = c->np / 16;
          lrc = launch_trace_terms(c);
          hipLaunchKernelGGL(k_sum_pairs_f64, dim3(1), dim3(256), 0, c->stream, c->dPart, nb, c->dScal);
          tt = c->dScal;
        }
        // the next launch of the chunk is a one-pass kernel: it evaluates this iteration's error and the
        // convergence test in its prologue (FusedCtl) -- no launch of its own for them
        const bool fold = fused && !c->fused8 && !h_only && j + 1 < chunk && c->algo != PMF_ALGO_SNMF;
        if (lrc == PMF_OK && fold) {
          c->conv_iter = i + j; c->conv_tt = tt; c->conv_ntt = ntt; c->conv_eps = conv_eps;
        } else if (lrc == PMF_OK) {
          hipLaunchKernelGGL(k_conv_check, dim3(1), dim3(64), 0, c->stream, tt, ntt, c->vnorm2, conv_eps,
                             (double)c->n, i + j, c->dFerr, c->dStop);
          if (hipGetLastError() != hipSuccess) lrc = fail(c, PMF_EHIP, "k_conv_check launch failed");
        }
        seq_after[j] = c->ipc_seq;
      }
      c->stop_arg = nullptr;
      c->conv_iter = -1;
      PMFCHK(lrc);
      int hstop[2] = {0, -1};
      HIPCHK(c, hipMemcpyAsync(hstop, c->dStop, sizeof(hstop), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(ferr + i, c->dFerr + i, (size_t)chunk * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      if (hstop[0] == 0) {                                // the whole chunk ran
        done += chunk;
        i += chunk - 1;
        continue;
      }
      // iterations i .. hstop[1] ran, the rest of the chunk were no-ops (with a communicator the
      // all-reduces still ran on the stale (P | S): it no longer belongs to W)
      const int s_it = hstop[1];
      done += s_it - i + 1;
      // The FOLDED exchanges behind the stop neither pushed nor waited (k_reduce_slabs_tiles / k_nmf_h_gram return on the
      // flag), on every rank alike (H, and with it the stop, is bit-identical across ranks): take their sequence numbers back,
      // so that the next exchange that really runs is the successor of the last one that did.  Counting the skipped ones
      // broke the two-slot invariant of pmf_ipc.h (a rank is at most one exchange ahead of a peer BECAUSE it needs that
      // peer's flags of exchange s + 1 before it can push s + 2 into the slot of s): after an odd number of skipped exchanges
      // the next push could land in a slot a slower peer was still adding up (round-5 advisor).  k_ipc_allreduce launches
      // run whatever the flag says, so a chunk that used those keeps its count.
      if (c->fold_calls - fold_calls0 == (long long)(c->ipc_seq - ipc_seq0) && c->ipc_seq != ipc_seq0) {
        const long long skipped = (long long)(c->ipc_seq - seq_after[s_it - i]);
        c->ipc_seq = seq_after[s_it - i];
        c->ipc_calls -= skipped; c->fold_calls -= skipped;
      }
      if (c->algo == PMF_ALGO_BNMF) {                     // only s_it - i + 1 H steps really ran
        c->lamb_w = lamb_w0; c->lamb_h = lamb_h0;
        for (int q = 0; q < s_it - i + 1; ++q) { c->lamb_w *= 1.1; c->lamb_h *= 1.1; }
      }
      if (c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_BNMF) {
        // the launches behind the stop were no-ops, but the host-side picture of where G lives was
        // advanced by them: put it back to what iteration s_it's H step (the last that ran) left
        const bool part = s_it + 1 < niter && !c->fused8;
        c->g_parts = part ? std::min(c->np / 64, PMF_HGRAM_MAX_WGS) : 0;
      }
      c->trace_ready = false;
      if (multi_rank(c)) { c->ps_valid = false; c->trace_ready = false; }
      if (hstop[0] == 1) {                                // nmf.py:198-202
        if (converged_at) *converged_at = s_it;
        break;
      }
      // the trace identity cancels at iteration s_it: evaluate it directly and go on in the
      // ordinary form, exactly what do_frobenius would have done
      free_run = false;
      i = s_it;
      PMFCHK(frobenius_direct(c, &ferr[i]));
    } else {
      if (gram) {
        PMFCHK(snmf_gram_iteration(c));                     // SNMF on k x n sized data (C = V^T V is at hand)
      } else if (cw && ch && c->algo == PMF_ALGO_SNMF && csr_fused_ok(c)) {
        PMFCHK(snmf_csr_fused_iteration(c));                // CSR: one pass over the rows
      } else if (fused) {                                   // update_w + update_h, one pass over V
        c->gram_partial_ok = (c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_BNMF) && i + 1 < niter && !c->fused8;
        PMFCHK(c->algo == PMF_ALGO_SNMF ? snmf_fused_iteration(c) : nmf_fused_iteration(c));
      } else {
        if (cw) PMFCHK(do_update_w(c));                     // nmf.py:183-184
        if (ch) PMFCHK(do_update_h(c));                     // nmf.py:186-187
      }
      ++done;
      if (ce && c->algo == PMF_ALGO_RNMF && ch) {           // update_s already summed (V - W H)^2
        HIPCHK(c, hipStreamSynchronize(c->stream));
        ferr[i] = std::sqrt(c->rnmf_err2);
      } else if (ce) {
        PMFCHK(do_frobenius(c, &ferr[i]));                  // nmf.py:189-190
      }
    }
    if (ce) {
      if (i > 1) {                                        // nmf.py:198
        const double derr = std::fabs(ferr[i] - ferr[i - 1]) / (double)c->n;   // nmf.py:135
        if (derr < conv_eps) {                            // nmf.py:136
          if (converged_at) *converged_at = i;            // caller: ferr = ferr[:i] (nmf.py:201)
          break;
        }
      }
    }
    if (can_free_run && !free_run && i + 1 >= kHostIters && niter - (i + 1) >= 2 && c->vnorm_valid &&
        ferr[i] * ferr[i] > 1e-2 * c->vnorm2) {
      // far from the cancellation threshold: hand the history to the device and let it run
      if (c->ferr_cap < niter) {
        if (c->dFerr) { HIPCHK(c, hipFree(c->dFerr)); c->dFerr = nullptr; }
        PMFCHK(dalloc(c, &c->dFerr, (size_t)niter));
        c->ferr_cap = niter;
      }
      if (!c->dStop) PMFCHK(dalloc(c, &c->dStop, 2));
      HIPCHK(c, hipMemcpyAsync(c->dFerr, ferr, (size_t)(i + 1) * sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemsetAsync(c->dStop, 0, 2 * sizeof(int), c->stream));
      free_run = true;
    }
  }
  c->want_trace = false;
  c->fixed_h_loop = false;
  c->gram_partial_ok = false;
  PMFCHK(materialize_w(c));      // Gram-space SNMF loop: W = V M once, inside the timed loop region
  HIPCHK(c, hipEventRecord(c->ev1, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  float ms = 0.f;
  HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
  c->last_loop_ms = ms;
  if (ce) for (int q = done; q < niter; ++q) ferr[q] = 0.0;   // as np.zeros(niter) leaves them (nmf.py:179-180)
  if (iters_done) *iters_done = done;
  wguard.ok = true;              // (materialize_w above joined the side stream)
  PMFCHK(ipc_check(c));
  return check_singular(c);
}

int pmf_set_lambda(pmf_ctx* c, double lamb_w, double lamb_h) {
  if (!c) return PMF_EINVAL;
  if (c->algo != PMF_ALGO_BNMF && c->algo != PMF_ALGO_RNMF)
    return fail(c, PMF_EINVAL, "pmf_set_lambda: only BNMF (penalty weights) and RNMF (threshold) take it");
  c->lamb_w = lamb_w; c->lamb_h = lamb_h;
  return PMF_OK;
}

int pmf_get_lambda(pmf_ctx* c, double* lamb_w, double* lamb_h) {
  if (!c || !lamb_w || !lamb_h) return PMF_EINVAL;
  *lamb_w = c->lamb_w; *lamb_h = c->lamb_h;
  return PMF_OK;
}

int pmf_rnmf_update_s(pmf_ctx* c) {
  PMFCHK(need(c, true, true, true));
  if (c->algo != PMF_ALGO_RNMF) return fail(c, PMF_EINVAL, "pmf_rnmf_update_s: RNMF only");
  PMFCHK(rnmf_update_s(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PMF_OK;
}

int pmf_rnmf_get_s_f32(pmf_ctx* c, float* S) {
  PMFCHK(need(c, true, false, false));
  if (c->algo != PMF_ALGO_RNMF || !c->s_valid || !S) return fail(c, PMF_EINVAL, "pmf_rnmf_get_s_f32: no S");
  // S = D + V on the device (the state kept is D = S - data, rnmf.py:102,111), then one download
  DevTemps tmp;
  float* dS = nullptr;
  PMFCHK(talloc(c, tmp, &dS, (size_t)c->mp * c->np));
  const int64_t E = c->mp * c->np;
  hipLaunchKernelGGL(k_add_f32, dim3(elem_grid(E / 4)), dim3(256), 0, c->stream, c->dD, c->dV, E, dS);
  HIPCHK(c, hipGetLastError());
  return download_padded(c, S, c->n, dS, c->np, c->m, c->n);
}

// The reference's S is an attribute: it travels with copies and pickles of the object (the host class hands it to the new
// context here) -- D = S - data, as pmf_rnmf_update_s leaves it.
int pmf_rnmf_set_s_f32(pmf_ctx* c, const float* S) {
  PMFCHK(need(c, true, false, false));
  if (c->algo != PMF_ALGO_RNMF || !S) return fail(c, PMF_EINVAL, "pmf_rnmf_set_s_f32: RNMF only, S must not be NULL");
  if (!c->dD) return fail(c, PMF_EINVAL, "pmf_rnmf_set_s_f32: no device state");
  PMFCHK(upload_padded(c, c->dD, c->np, S, c->n, c->m, c->n));
  c->s_valid = true;
  PMFCHK(rnmf_data_change_end(c));                 // D = S - V
  c->ps_valid = false; c->trace_ready = false;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PMF_OK;
}

// ---- streamed V: one pass = one reference iteration over row tiles handed in by the caller ------
// (the `data[:, :]` idiom of nmf.py:123,129 for data that does not fit in HBM --
// an h5py dataset, a memmap, a matrix larger than 288 GB.)  W stays resident; a tile is visited once
// per pass: W step on its rows (nmf.py:128-132), then the partials of W^T V and W^T W of the NEW
// rows, accumulated in float64 over the tiles; pmf_stream_end all-reduces them, runs the H step
// (nmf.py:122-126) and evaluates ||V - W H|| by the trace identity.  Copies run on their own
// stream into two device tiles, so the copy of tile t+1 overlaps the kernels of tile t.
int pmf_stream_begin(pmf_ctx* c, uint32_t flags, int64_t max_tile_rows) {
  if (c) c->hd_synced = false;
  if (!c) return PMF_EINVAL;
  if (c->algo == PMF_ALGO_RNMF)   // (the reference's RNMF keeps S, an in-memory array of data's shape: rnmf.py:94-98)
    return fail(c, PMF_EINVAL, "pmf_stream_*: NMF, BNMF, SNMF and NMFALS contexts");
  if (!c->have_w || !c->have_h) return fail(c, PMF_EINVAL, "pmf_stream_begin: W and H must be set");
  if (max_tile_rows < 1) return fail(c, PMF_EINVAL, "pmf_stream_begin: max_tile_rows must be >= 1");
  HIPCHK(c, hipSetDevice(c->device));
  const int64_t cap = std::min<int64_t>(round_up(max_tile_rows, 64), c->mp);
  if (!c->copy_stream) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_copied[b], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_consumed[b], hipEventDisableTiming));
    }
    PMFCHK(dalloc(c, &c->dPSacc, (size_t)ps_elems(c)));
    PMFCHK(dalloc(c, &c->dStAcc, 4));
  }
  if (cap > c->tile_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    for (int b = 0; b < 2; ++b) {
      if (c->dTile[b]) { HIPCHK(c, hipFree(c->dTile[b])); c->dTile[b] = nullptr; }
      PMFCHK(dalloc(c, &c->dTile[b], (size_t)cap * c->np));    // zeroed: pad columns stay 0
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->tile_cap = cap;
  }
  c->st_flags = flags;
  c->st_rows_seen = 0;
  c->st_tiles = 0;
  c->st_active = true;
  c->st_vnorm_pending = !(flags & PMF_STREAM_RESID) && !c->vnorm_valid;
  if ((flags & PMF_COMPUTE_W) && !(flags & PMF_STREAM_RESID)) {
    // what the W step of every tile needs from H: G = H H^T (NMF, BNMF); M^T = inv(H H^T) H (SNMF, snmf.py:67-70);
    // the Hessian H H^T in float64 + the warm-start verdict (NMFALS, nmfals.py:85-97)
    if (c->algo == PMF_ALGO_SNMF) PMFCHK(snmf_inverse(c));
    else if (c->algo == PMF_ALGO_NMFALS) { PMFCHK(ensure_gram(c, 1.0)); PMFCHK(nnqp_prepare(c, c->stream, nnqp_use_wave(c))); }
    else PMFCHK(ensure_gram(c, 0.0));
  }
  return PMF_OK;
}

int pmf_stream_tile(pmf_ctx* c, int64_t row0, int64_t rows, const float* tile, int64_t ld) {
  if (!c) return PMF_EINVAL;
  if (!c->st_active) return fail(c, PMF_EINVAL, "pmf_stream_tile: no pass open (pmf_stream_begin)");
  if (!tile || ld < c->n || rows < 1 || rows > c->tile_cap || row0 != c->st_rows_seen || row0 + rows > c->m ||
      (row0 % 64) != 0 || (row0 + rows < c->m && (rows % 64) != 0))
    return fail(c, PMF_EINVAL, "pmf_stream_tile: tiles must arrive in row order, start on a multiple of 64 rows, "
                               "hold a multiple of 64 rows (except the last) and fit max_tile_rows");
  HIPCHK(c, hipSetDevice(c->device));
  const int b = c->st_tiles & 1;
  const int64_t rows_p = round_up(rows, 64);
  float* T = c->dTile[b];
  if (c->st_tiles >= 2) {
    // The copy below is enqueued BEHIND the consumer of this buffer's previous tile and reads the caller's memory when it
    // runs, not now: without a bound the host gets many tiles ahead of the device and a caller that hands over temporaries
    // (a float64 or strided `data` converted tile by tile) frees them long before they are read -- garbage tiles, NaN factors
    // (found by tests/sweeps/fuzz_sequences.py).  Waiting here for the COPY of the tile two calls back bounds the host's lead
    // to the two device tiles: a caller's tile must stay valid until the second-next call (or pmf_stream_end) has returned.
    HIPCHK(c, hipEventSynchronize(c->ev_copied[b]));
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_consumed[b], 0));
  }
  HIPCHK(c, hipMemcpy2DAsync(T, (size_t)c->np * sizeof(float), tile, (size_t)ld * sizeof(float),
                             (size_t)c->n * sizeof(float), (size_t)rows, hipMemcpyHostToDevice, c->copy_stream));
  if (rows_p > rows)
    HIPCHK(c, hipMemsetAsync(T + rows * c->np, 0, (size_t)(rows_p - rows) * c->np * sizeof(float), c->copy_stream));
  HIPCHK(c, hipEventRecord(c->ev_copied[b], c->copy_stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copied[b], 0));
  float* Wt = c->dW + row0 * c->KP;
  const int first = c->st_tiles == 0;
  if (c->st_flags & PMF_STREAM_RESID) {
    if (c->nb > 1) {            // num_bases > 128: launch_resid's kernels end at 128 bases (found by tests/sweeps/fuzz_sequences.py)
      PMFCHK(resid_bigk(c, false, c->dScal + 5, T, Wt, rows_p));
    } else {
      PMFCHK(launch_resid(c, false, 0.f, T, Wt, rows_p));
      hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(256), 0, c->stream, c->dPart, c->resid_parts, c->dScal + 5);
      HIPCHK(c, hipGetLastError());
    }
    hipLaunchKernelGGL(k_accum_f64, dim3(1), dim3(64), 0, c->stream, c->dStAcc + 1, c->dScal + 5, first);
    HIPCHK(c, hipGetLastError());
  } else {
    if (c->st_vnorm_pending) {
      const int nb = 256;
      hipLaunchKernelGGL(k_sumsq, dim3(nb), dim3(256), 0, c->stream, T, rows_p * c->np, c->dPart);
      HIPCHK(c, hipGetLastError());
      hipLaunchKernelGGL(k_sum_f64, dim3(1), dim3(256), 0, c->stream, c->dPart, nb, c->dScal + 5);
      HIPCHK(c, hipGetLastError());
      hipLaunchKernelGGL(k_accum_f64, dim3(1), dim3(64), 0, c->stream, c->dStAcc, c->dScal + 5, first);
      HIPCHK(c, hipGetLastError());
    }
    if (c->st_flags & PMF_COMPUTE_W) {
      if (c->nb > 1 && (c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_BNMF)) {   // blocks of 128 bases (bigk_update_w)
        PMFCHK(bigk_update_w_rows(c, T, Wt, c->dW1 + row0 * c->KP, c->dW2 + row0 * c->KP, rows_p, rows));
      } else if ((c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_BNMF) && c->np > PMF_WIDE_K) {   // (very wide data)
        PMFCHK(wide_update_w_rows(c, T, Wt, rows_p, rows));
      } else if (c->algo == PMF_ALGO_BNMF) {   // bnmf.py:87-90: the penalised W rule, same contractions
        PMFCHK(rowgemm<EPI_BNMF_W>(c, T, c->np, c->np, c->dH, c->np, Wt, c->dG, nullptr, rows_p, rows));
      } else if (c->algo == PMF_ALGO_SNMF) {   // snmf.py:67-70: the tile's rows of W = V M^T
        PMFCHK(rowgemm<EPI_STORE>(c, T, c->np, c->np, c->dMT, c->np, nullptr, nullptr, Wt, rows_p, rows));
      } else if (c->algo == PMF_ALGO_NMFALS) {  // nmfals.py:85-97: right-hand sides V H^T of the tile's rows, one QP per row
        float* Ft = c->dW1 + row0 * c->KP;
        PMFCHK(rowgemm<EPI_STORE>(c, T, c->np, c->np, c->dH, c->np, nullptr, nullptr, Ft, rows_p, rows));
        if (nnqp_use_wave(c)) {                  // 64 < num_bases <= 128: k_nnqp_wave, inv(HA) prepared by pmf_stream_begin
          PMFCHK(solve_nnqps(c, Ft, 1, c->KP, Wt, 1, c->KP, rows, false, true));
        } else {
          double* qp = nullptr;
          PMFCHK(nnqp_scratch(c, &qp));
          const int qrc = pmf_launch_nnqp(c->stream, c->KP, c->k, c->dGd, Ft, 1, c->KP, Wt, 1, c->KP, rows, c->dWarm, qp, 0);
          if (qrc != PMF_OK) return fail(c, qrc, "nnqp launch (streamed W tile) failed");
          HIPCHK(c, hipGetLastError());
        }
      } else {
        PMFCHK(rowgemm<EPI_NMF_W>(c, T, c->np, c->np, c->dH, c->np, Wt, c->dG, nullptr, rows_p, rows));
      }
      c->ps_valid = false; c->num_valid = false; c->trace_ready = false;
    }
    if ((c->st_flags & (PMF_COMPUTE_H | PMF_COMPUTE_ERR)) && !((c->st_flags & PMF_COMPUTE_W) == 0 && c->ps_valid)) {
      const int64_t blocks16 = rows_p / 16;
      int tch = (int)std::min<int64_t>(c->nchunks, blocks16);
      const int rpc = (int)round_up((blocks16 + tch - 1) / tch, 4) * 16;
      tch = (int)((rows_p + rpc - 1) / rpc);
      if (c->nb > 1) {
        PMFCHK(bigk_ps_rows(c, T, Wt, rows_p, rpc, tch, c->dPSacc, first));
      } else {
      PMFCHK(colgemm_rows(c, T, Wt, rows_p, rpc, tch));
      const int64_t E = ps_elems(c);
      hipLaunchKernelGGL(k_reduce_slabs_acc, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, c->stream, c->dSlab, tch,
                         E, c->dPSacc, first);
      HIPCHK(c, hipGetLastError());
      }
    }
  }
  HIPCHK(c, hipEventRecord(c->ev_consumed[b], c->stream));
  c->st_rows_seen += rows;
  c->st_tiles += 1;
  return PMF_OK;
}

int pmf_stream_end(pmf_ctx* c, double* ferr, int32_t* needs_direct) {
  if (!c) return PMF_EINVAL;
  if (!c->st_active) return fail(c, PMF_EINVAL, "pmf_stream_end: no pass open");
  c->st_active = false;
  c->hd_synced = false;
  if (needs_direct) *needs_direct = 0;
  if (c->st_rows_seen != c->m)
    return fail(c, PMF_EINVAL, "pmf_stream_end: the tiles covered " + std::to_string(c->st_rows_seen) + " of " +
                std::to_string(c->m) + " rows");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->st_flags & PMF_STREAM_RESID) {
    PMFCHK(allreduce_sum(c, c->dStAcc + 1, 1, true));
    double ss = 0.0;
    HIPCHK(c, hipMemcpyAsync(&ss, c->dStAcc + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ferr) *ferr = std::sqrt(ss);
    return PMF_OK;
  }
  if (c->st_vnorm_pending) {
    PMFCHK(allreduce_sum(c, c->dStAcc, 1, true));
    HIPCHK(c, hipMemcpyAsync(&c->vnorm2, c->dStAcc, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->vnorm_valid = true;
    c->st_vnorm_pending = false;
  }
  const bool need_ps = (c->st_flags & (PMF_COMPUTE_H | PMF_COMPUTE_ERR)) != 0;
  if (need_ps && !c->ps_valid) {
    const int64_t E = ps_elems(c);
    hipLaunchKernelGGL(k_f64_to_f32, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, c->stream, c->dPSacc, E, c->dPS);
    HIPCHK(c, hipGetLastError());
    PMFCHK(allreduce_ps(c));
    c->ps_valid = true;
    c->trace_ready = false;       // (terms an earlier H step left belong to an earlier (P | S))
  }
  if (c->st_flags & PMF_COMPUTE_H) {
    c->want_trace = (c->st_flags & PMF_COMPUTE_ERR) != 0;
    const int rc = c->algo == PMF_ALGO_NMFALS ? als_update_h(c) : h_step_from_ps(c);   // nmfals.py:70-82: QPs over the summed (P | S)
    c->want_trace = false;
    PMFCHK(rc);
  }
  if ((c->st_flags & PMF_COMPUTE_ERR) && ferr) {
    double e2 = 0.0;
    PMFCHK(trace_e2(c, &e2));
    if (!(e2 > 1e-3 * c->vnorm2) && needs_direct) *needs_direct = 1;   // cancellation: ask for a PMF_STREAM_RESID pass
    *ferr = std::sqrt(e2 > 0.0 ? e2 : 0.0);
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return check_singular(c);
}

int pmf_nndsvd_init(pmf_ctx* c, int32_t* rank_found) {
  PMFCHK(need(c, true, false, false));
  if (c->algo == PMF_ALGO_RNMF) return fail(c, PMF_EINVAL, "pmf_nndsvd_init: not for RNMF contexts");
  return nndsvd_init(c, rank_found);
}


int pmf_nnqp_counters(pmf_ctx* c, int64_t* out8, int32_t reset) {
  if (!c || !out8) return PMF_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c->dQstat) {
    HIPCHK(c, hipMemcpyAsync(h, c->dQstat, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    if (reset) HIPCHK(c, hipMemsetAsync(c->dQstat, 0, sizeof(h), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  for (int q = 0; q < 8; ++q) out8[q] = (int64_t)h[q];
  return PMF_OK;
}

int pmf_last_loop_ms(pmf_ctx* c, double* ms) {
  if (!c || !ms) return PMF_EINVAL;
  *ms = c->last_loop_ms;
  return PMF_OK;
}

int pmf_collective_ms(pmf_ctx* c, double* mean_ms, int64_t* count) {
  if (!c || !mean_ms || !count) return PMF_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double sum = 0.0;
  int64_t n = 0;
  for (size_t q = 0; q + 1 < c->coll_used; q += 2) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->coll_ev[q], c->coll_ev[q + 1]) == hipSuccess) { sum += ms; ++n; }
  }
  if (c->dIpcWait) {
    // the folded exchanges have no launch to bracket: what they cost an iteration is the consumer's wait for the slowest
    // peer's flags in k_nmf_h_gram's prologue (ticks of the 100 MHz counter, summed on the device since pmf_profile_enable)
    unsigned long long w[2] = {0, 0};
    HIPCHK(c, hipMemcpy(w, c->dIpcWait, sizeof(w), hipMemcpyDeviceToHost));
    sum += (double)w[0] * 1e-5;        // 100 MHz ticks -> ms
    n += (int64_t)w[1];
  }
  *mean_ms = n ? sum / (double)n : 0.0;
  *count = n;
  return PMF_OK;
}

int pmf_profile_enable(pmf_ctx* c, int32_t on) {
  if (!c) return PMF_EINVAL;
  c->profile = on != 0;
  c->stat.used = 0;
  c->stat.seen = 0;
  c->stat.open = false;
  c->coll_seen = 0;
  c->coll_used = 0;
  if (c->dIpcWait) HIPCHK(c, hipMemsetAsync(c->dIpcWait, 0, 2 * sizeof(unsigned long long), c->stream));
  return PMF_OK;
}

int pmf_kernel_stats(pmf_ctx* c, const char** name, int64_t* launches, double* mean_ms,
                     double* flops_per_launch, double* bytes_per_launch) {
  if (!c) return PMF_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  double tot = 0.0;
  const size_t pairs = c->stat.used / 2;
  for (size_t q = 0; q < pairs; ++q) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->stat.ev[2 * q], c->stat.ev[2 * q + 1]));
    tot += ms;
  }
  if (name) *name = c->stat.name.c_str();
  if (launches) *launches = (int64_t)pairs;
  if (mean_ms) *mean_ms = pairs ? tot / (double)pairs : 0.0;
  if (flops_per_launch) *flops_per_launch = c->stat.flops;
  if (bytes_per_launch) *bytes_per_launch = c->stat.bytes;
  return PMF_OK;
}

// ---- host-side change detector for the caller's arrays (no device involved) -------------------
// The reference computes from whatever self.W / self.H / self.data hold at the time of the call
// (nmf.py:122-132); the host class keeps device copies and must notice ANY in-place edit of the host
// arrays.  A sum misses permutations; this is an order-dependent 128-bit digest of the raw bytes:
// 8 interleaved multiply-rotate lanes per 64-byte line (memory speed), 8 MiB chunks hashed by up to
// 16 threads and folded in chunk order.
static inline uint64_t pmf_rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t pmf_fmix64(uint64_t h) {
  h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
  return h;
}
static void pmf_hash_chunk(const unsigned char* p, size_t n, uint64_t seed, uint64_t out[2]) {
  constexpr uint64_t P1 = 0x9e3779b185ebca87ull, P2 = 0xc2b2ae3d27d4eb4full;
  uint64_t a[8];
  for (int l = 0; l < 8; ++l) a[l] = pmf_fmix64(seed + P1 * (uint64_t)(l + 1));
  const size_t lines = n / 64;
  for (size_t i = 0; i < lines; ++i) {
    uint64_t w[8];
    std::memcpy(w, p + 64 * i, 64);
    for (int l = 0; l < 8; ++l) a[l] = (pmf_rotl64(a[l], 31) ^ w[l]) * P1;
  }
  if (n % 64) {
    uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::memcpy(w, p + 64 * lines, n % 64);
    for (int l = 0; l < 8; ++l) a[l] = (pmf_rotl64(a[l], 31) ^ w[l]) * P1;
  }
  uint64_t h1 = (uint64_t)n * P2, h2 = seed ^ P2;
  for (int l = 0; l < 8; ++l) {
    h1 = pmf_rotl64(h1, 27) * P1 + pmf_fmix64(a[l]);
    h2 = (pmf_rotl64(h2, 29) ^ pmf_fmix64(a[l] + P2 * (uint64_t)(l + 1))) * P2;
  }
  out[0] = pmf_fmix64(h1);
  out[1] = pmf_fmix64(h2);
}

int pmf_host_checksum(const void* data, uint64_t nbytes, uint64_t* out2) {
  if (!out2 || (nbytes && !data)) return PMF_EINVAL;
  const unsigned char* p = static_cast<const unsigned char*>(data);
  constexpr size_t CH = (size_t)8 << 20;
  const size_t nch = nbytes ? (size_t)((nbytes + CH - 1) / CH) : 1;
  std::vector<uint64_t> dig(2 * nch);
  auto work = [&](size_t c0, size_t c1) {
    for (size_t q = c0; q < c1; ++q) {
      const size_t off = q * CH;
      pmf_hash_chunk(p + off, (size_t)std::min<uint64_t>(CH, nbytes - off), 0x243f6a8885a308d3ull + q, &dig[2 * q]);
    }
  };
  unsigned nthr = std::min<unsigned>(std::min<size_t>(nch, 16), std::max(1u, std::thread::hardware_concurrency()));
  if (nthr <= 1) {
    work(0, nch);
  } else {
    std::vector<std::thread> th;
    bool spawned_all = true;
    size_t next = 0;
    try {
      for (unsigned t = 0; t < nthr; ++t) {
        const size_t c0 = nch * t / nthr, c1 = nch * (t + 1) / nthr;
        th.emplace_back(work, c0, c1);
        next = c1;
      }
    } catch (...) { spawned_all = false; }
    for (auto& t : th) t.join();
    if (!spawned_all) work(next, nch);
  }
  uint64_t h1 = 0x13198a2e03707344ull ^ nbytes, h2 = 0xa4093822299f31d0ull;
  for (size_t q = 0; q < nch; ++q) {
    h1 = pmf_fmix64(pmf_rotl64(h1, 23) * 0x9e3779b185ebca87ull + dig[2 * q]);
    h2 = pmf_fmix64((pmf_rotl64(h2, 37) ^ dig[2 * q + 1]) * 0xc2b2ae3d27d4eb4full);
  }
  out2[0] = h1;
  out2[1] = h2;
  return PMF_OK;
}

// The caller changed V behind the library's back (a streamed `data` object was rebound or edited): forget
// everything derived from it -- ||V||^2, (W^T V | W^T W), the cached V H^T.
int pmf_set_option(pmf_ctx* c, const char* name, int64_t value) {
  if (!c || !name) return PMF_EINVAL;
  if (std::strcmp(name, "force_tiled") == 0) {
    if (value != 0 && value != 1) return fail(c, PMF_EINVAL, "force_tiled: 0 or 1");
    if (value == 1 && c->fused_wgs > 0) {
      c->fused_wgs_hidden = c->fused_wgs; c->fused8_hidden = c->fused8;
      c->fused_wgs = 0; c->fused8 = false;
      c->path_hidden = c->path;
      c->path = "tiled (forced)";
    } else if (value == 0 && c->fused_wgs_hidden > 0) {
      c->fused_wgs = c->fused_wgs_hidden; c->fused8 = c->fused8_hidden;
      c->fused_wgs_hidden = 0;
      c->path = c->path_hidden;
    }
    c->ps_valid = false; c->num_valid = false; c->trace_ready = false;
    if (c->g_parts > 0) { c->g_valid = false; c->g_parts = 0; }
    choose_stat_site(c, false);
    return PMF_OK;
  }
  if (std::strcmp(name, "profile_every") == 0) {
    if (value < 1 || value > 1 << 20) return fail(c, PMF_EINVAL, "profile_every: 1 .. 2^20");
    c->stat.every = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "fold_exchange") == 0) {
    if (value != 0 && value != 1) return fail(c, PMF_EINVAL, "fold_exchange: 0 or 1");
    c->opt_fold = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "oneshot_allreduce") == 0) {
    if (value != 0 && value != 1) return fail(c, PMF_EINVAL, "oneshot_allreduce: 0 or 1");
    if (value == 1 && c->ipc_nranks_ready <= 1) return fail(c, PMF_EINVAL, "oneshot_allreduce: not set up (pmf_ipc_export / pmf_ipc_import)");
    c->ipc.nranks = value ? c->ipc_nranks_ready : 0;
    return PMF_OK;
  }
  if (std::strcmp(name, "nnqp_wave") == 0) {
    if (value != 0 && value != 1) return fail(c, PMF_EINVAL, "nnqp_wave: 0 or 1");
    c->opt_nnqp_wave = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "snmf_w_pipe") == 0) {
    if (value < 0 || value > 256) return fail(c, PMF_EINVAL, "snmf_w_pipe: 0 (off) .. 256 workgroup slots left free by the W write");
    HIPCHK(c, hipSetDevice(c->device));
    PMFCHK(w_pipe_join(c));
    c->opt_w_pipe = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "nnqp_count") == 0) {
    if (value != 0 && value != 1) return fail(c, PMF_EINVAL, "nnqp_count: 0 or 1");
    c->opt_nnqp_count = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "nnqp_frame16") == 0) {
    if (value != 0 && value != 1) return fail(c, PMF_EINVAL, "nnqp_frame16: 0 or 1");
    c->opt_nnqp_frame16 = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "nnqp_quad") == 0) {
    if (value < 0 || value > 2) return fail(c, PMF_EINVAL, "nnqp_quad: 0 (never), 1 (from 16 384 problems per half step on) or 2 (always)");
    c->opt_nnqp_quad = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "nndsvd_topk") == 0) {
    if (value < -1 || value > 1) return fail(c, PMF_EINVAL, "nndsvd_topk: -1 (by size), 0 or 1");
    c->opt_nndsvd_topk = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "colgemm_stream") == 0) {
    if (value != 0 && value != 1) return fail(c, PMF_EINVAL, "colgemm_stream: 0 or 1");
    c->opt_colgemm_stream = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "rowgemm_stream") == 0) {
    if (value != 0 && value != 1) return fail(c, PMF_EINVAL, "rowgemm_stream: 0 or 1");
    c->opt_rowgemm_stream = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "fuse_chain") == 0) {
    if (value < 0 || value > 3) return fail(c, PMF_EINVAL, "fuse_chain: bit 0 = the W half step's Gram + inverse, bit 1 = the H half step's slab sum + inverse");
    c->opt_fuse_chain = (int)value;
    return PMF_OK;
  }
  if (std::strcmp(name, "snmf_h64") == 0) {
    if (value != 0 && value != 1) return fail(c, PMF_EINVAL, "snmf_h64: 0 (H in float32 between the steps, rounds 1-5) or 1 (H in float64 on the device)");
    c->opt_snmf_h64 = (int)value;
    c->hd_synced = false; c->g_valid = false; c->g_parts = 0;
    return PMF_OK;
  }
  if (std::strcmp(name, "snmf_gram") == 0) {
    if (value < -1 || value > 2) return fail(c, PMF_EINVAL, "snmf_gram: -1 (auto), 0 (off), 1 (on) or 2 (on, W written every iteration)");
    c->opt_snmf_gram = (int)value;
    return PMF_OK;
  }
  return fail(c, PMF_EINVAL, std::string("pmf_set_option: unknown option '") + name + "'");
}

int pmf_set_host_allreduce(pmf_ctx* c, pmf_host_allreduce_fn fn, void* user) {
  if (!c) return PMF_EINVAL;
  if (fn && c->comm) return fail(c, PMF_EINVAL, "pmf_set_host_allreduce: the context already has an RCCL communicator");
  c->host_ar = fn;
  c->host_ar_user = user;
  c->ps_valid = false; c->vnorm_valid = false; c->trace_ready = false; c->c_valid = false;
  return PMF_OK;
}

// ---- one-shot all-reduce over IPC-mapped receive areas (pmf_ipc.h) ----
int pmf_ipc_export(pmf_ctx* c, int32_t rank, int32_t nranks, void* handle_out) {
  static_assert(sizeof(hipIpcMemHandle_t) <= PMF_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
  if (!c || !handle_out || nranks < 2 || nranks > PMF_IPC_MAX_RANKS || rank < 0 || rank >= nranks)
    return fail(c, PMF_EINVAL, "pmf_ipc_export: 2 <= nranks <= 8, 0 <= rank < nranks");
  if (c->ipc_exported) return fail(c, PMF_EINVAL, "pmf_ipc_export: already exported");
  HIPCHK(c, hipSetDevice(c->device));
  void* area = nullptr;
  const size_t bytes = pmf_ipc_area_bytes(nranks);
  // fine-grained device memory: stores of a peer on another GPU must become visible WHILE this rank's kernel polls
  hipError_t e = hipExtMallocWithFlags(&area, bytes, hipDeviceMallocFinegrained);
  hipIpcMemHandle_t h;
  if (e == hipSuccess && hipIpcGetMemHandle(&h, area) != hipSuccess) { (void)hipFree(area); area = nullptr; e = hipErrorUnknown; }
  if (e != hipSuccess) {                       // (a runtime that cannot export fine-grained memory: ordinary device memory)
    (void)hipGetLastError();
    // ... which is only sound between processes on ONE device: coarse-grained memory is, by the memory model, touched by a single
    // agent while a kernel runs -- a peer GPU's stores would meet stale L2 lines here.  With an RCCL communicator (ranks on GPUs of
    // their own) the export fails instead and RCCL keeps carrying the sums.
    if (c->comm != nullptr)
      return fail(c, PMF_EHIP, "pmf_ipc_export: fine-grained device memory cannot be allocated / exported on this runtime");
    HIPCHK(c, hipMalloc(&area, bytes));
    HIPCHK(c, hipIpcGetMemHandle(&h, area));
  }
  HIPCHK(c, hipMemsetAsync(area, 0, bytes, c->stream));
  if (!c->dIpcErr) PMFCHK(dalloc(c, &c->dIpcErr, 1));
  if (!c->dIpcWait) {
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->dIpcWait), 2 * sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(c->dIpcWait, 0, 2 * sizeof(unsigned long long), c->stream));
  }
  // the self-test's two payload buffers, here: everything that can fail on ONE rank alone happens before the ranks vote on
  // "exported"; between the self-test's collectives nothing is allocated
  if (!c->dIpcTestA) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->dIpcTestA), PMF_IPC_MAX_BYTES));
  if (!c->dIpcTestB) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->dIpcTestB), PMF_IPC_MAX_BYTES));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memset(handle_out, 0, PMF_IPC_HANDLE_BYTES);
  std::memcpy(handle_out, &h, sizeof(h));
  c->ipc = IpcPeers{};
  c->ipc.area[rank] = static_cast<char*>(area);
  c->ipc.me = rank;
  c->ipc.nranks = 0;                          // ready only after pmf_ipc_import
  c->ipc_exported = true;
  c->ipc_export_nranks = nranks;              // the receive area is sized for THIS many ranks
  c->ipc_seq = 0;
  return PMF_OK;
}

int pmf_ipc_import(pmf_ctx* c, const void* handles, int32_t nranks) {
  if (!c || !handles || !c->ipc_exported || nranks != c->ipc_export_nranks || c->ipc_nranks_ready > 0)
    return fail(c, PMF_EINVAL, "pmf_ipc_import: call pmf_ipc_export first, once; handles = nranks x PMF_IPC_HANDLE_BYTES in rank "
                               "order with the nranks given to pmf_ipc_export (the receive areas are sized for it)");
  HIPCHK(c, hipSetDevice(c->device));
  for (int r = 0; r < nranks; ++r) {
    if (r == c->ipc.me) continue;
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const char*>(handles) + (size_t)r * PMF_IPC_HANDLE_BYTES, sizeof(h));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {                     // give back what was mapped so far: nothing half-open survives a failed import
      (void)hipGetLastError();
      for (int q = 0; q < nranks; ++q)
        if (q != c->ipc.me && c->ipc.area[q]) { (void)hipIpcCloseMemHandle(c->ipc.area[q]); c->ipc.area[q] = nullptr; }
      return fail(c, PMF_EHIP, std::string("pmf_ipc_import: hipIpcOpenMemHandle of rank ") + std::to_string(r) + ": " + hipGetErrorString(e));
    }
    c->ipc.area[r] = static_cast<char*>(p);
  }
  c->ipc.nranks = nranks;
  c->ipc_nranks_ready = nranks;
  return PMF_OK;
}

// The one-shot path against the context's other transport (RCCL or the host callback) on rank- and round-dependent
// payloads of the per-iteration size; several rounds, so that both slots are reused.  *ok = 1 iff every round agreed and
// no wait ran out.  The caller combines the ranks' verdicts and switches the path off everywhere when any rank saw a
// difference (pmf_set_option("oneshot_allreduce", 0)): an all-reduce that has never run on the machine at hand does not
// get to carry the iteration on trust.
int pmf_ipc_selftest(pmf_ctx* c, int32_t rounds, int32_t* ok) {
  if (!c || !ok) return PMF_EINVAL;
  *ok = 0;
  if (c->ipc_nranks_ready <= 1) return fail(c, PMF_EINVAL, "pmf_ipc_selftest: no one-shot all-reduce set up (pmf_ipc_import)");
  if (!c->comm && !c->host_ar) return fail(c, PMF_EINVAL, "pmf_ipc_selftest: needs a second transport to compare with");
  if (!c->dIpcTestA || !c->dIpcTestB) return fail(c, PMF_EINVAL, "pmf_ipc_selftest: no test buffers (pmf_ipc_export allocates them)");
  const size_t count = std::min<size_t>((size_t)ps_elems(c), PMF_IPC_MAX_BYTES / sizeof(float));
  float *dA = c->dIpcTestA, *dB = c->dIpcTestB;
  std::vector<float> x(count), a(count), b(count);
  bool good = hipSetDevice(c->device) == hipSuccess;
  int rc = PMF_OK;
  c->ipc_wait_ticks = 2ull * 100000000ull;             // 2 s: the ranks enter the test together
  // EVERY rank runs EVERY round whatever it has seen so far: a rank that left early would leave its peers waiting in the
  // next round's collectives
  if (rounds < 2) rounds = 2;                          // (odd rounds run the split form the loop uses: never skipped)
  // the split form at THIS context's sizes: one flag per (P | S) tile of k_reduce_slabs_tiles, polled by every workgroup of
  // a consumer grid as large as launch_h_gram's
  const int ntu_ctx = c->NT * (c->np / 16) + c->NT * (c->NT + 1) / 2;
  const int nfl = ntu_ctx >= 1 && ntu_ctx <= PMF_IPC_MAX_WGS ? ntu_ctx : 74;   // (74: 64 bases x 256 columns)
  const int npull = std::max(1, std::min(c->np / 64, PMF_HGRAM_MAX_WGS));
  for (int t = 0; t < rounds; ++t) {
    for (size_t i = 0; i < count; ++i) x[i] = (float)((c->ipc.me + 1) * (t + 1)) + 0.001f * (float)(i % 977);
    const size_t cnt = t % 3 == 2 ? std::max<size_t>(1, count / 3) : count;      // (a shorter payload now and then)
    if (hipMemcpyAsync(dA, x.data(), cnt * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipMemcpyAsync(dB, x.data(), cnt * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = PMF_EHIP; good = false; }
    c->ipc.nranks = c->ipc_nranks_ready;
    if (t & 1) {
      // the SPLIT form the loop uses (push inside a producer launch of many workgroups, wait + rank-ordered sum inside a
      // consumer launch): same slots, flags and sequence counter
      const unsigned seq = ++c->ipc_seq;
      hipLaunchKernelGGL(k_ipc_fold_push, dim3(nfl), dim3(256), 0, c->stream, dA, (int64_t)cnt, c->ipc, seq);
      hipLaunchKernelGGL(k_ipc_fold_pull, dim3(npull), dim3(1024), 0, c->stream, dA, (int64_t)cnt, c->ipc, seq, nfl, c->dIpcErr, c->ipc_wait_ticks);
      if (hipGetLastError() != hipSuccess) { rc = PMF_EHIP; good = false; }
      ++c->ipc_calls;
    } else if (allreduce_sum(c, dA, cnt, false) != PMF_OK) { rc = PMF_EHIP; good = false; }
    c->ipc.nranks = 0;                                 // the other transport
    if (allreduce_sum(c, dB, cnt, false) != PMF_OK) { rc = PMF_EHIP; good = false; }
    c->ipc.nranks = c->ipc_nranks_ready;
    if (hipMemcpyAsync(a.data(), dA, cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(b.data(), dB, cnt * sizeof(float), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { rc = PMF_EHIP; good = false; continue; }
    if (ipc_check(c) != PMF_OK) good = false;
    for (size_t i = 0; i < cnt && good; ++i)
      if (!(std::fabs(a[i] - b[i]) <= 1e-5f * std::fabs(b[i]))) good = false;
  }
  c->ipc_wait_ticks = PMF_IPC_WAIT_TICKS;
  (void)rc;                                            // (a failing call is a failed test, not an error of this function)
  *ok = good ? 1 : 0;
  return PMF_OK;
}

const char* pmf_collective_name(pmf_ctx* c) {
  if (!c) return "";
  static thread_local std::string s;
  const bool ipc = c->ipc.nranks > 1;
  s = ipc ? "one-shot IPC all-reduce (payloads <= 256 KiB: every rank writes into every peer's receive area, sums in rank order)" : "";
  if (c->comm) s += std::string(ipc ? " + " : "") + "ncclAllReduce (RCCL)" + (ipc ? " for larger payloads" : "");
  if (c->host_ar) s += std::string(s.empty() ? "" : " + ") + "host transport (pmf_set_host_allreduce)" + (ipc ? " for larger payloads" : "");
  if (s.empty()) s = "none";
  char buf[160];
  snprintf(buf, sizeof(buf), " [calls: ipc %lld (%lld of them folded into the slab-reduce / H-step launches), rccl %lld, host %lld]",
           (long long)c->ipc_calls, (long long)c->fold_calls, (long long)c->rccl_calls, (long long)c->host_calls);
  s += buf;
  return s.c_str();
}

int pmf_invalidate_v(pmf_ctx* c) {
  if (!c) return PMF_EINVAL;
  c->vnorm_valid = false; c->vnorm_local_valid = false; c->ps_valid = false; c->num_valid = false; c->trace_ready = false;
  c->c_valid = false;
  return PMF_OK;
}

int pmf_snapshot_w(pmf_ctx* c) {
  PMFCHK(need(c, false, true, false));
  PMFCHK(materialize_w(c));
  if (!c->dWsnap) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->dWsnap), (size_t)c->mp * c->KP * sizeof(float)));
  HIPCHK(c, hipMemcpyAsync(c->dWsnap, c->dW, (size_t)c->mp * c->KP * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  c->wsnap_valid = true;
  return PMF_OK;
}

int pmf_restore_w(pmf_ctx* c) {
  if (!c) return PMF_EINVAL;
  if (!c->wsnap_valid) return fail(c, PMF_EINVAL, "pmf_restore_w: no snapshot (pmf_snapshot_w)");
  HIPCHK(c, hipSetDevice(c->device));
  PMFCHK(w_pipe_join(c));
  HIPCHK(c, hipMemcpyAsync(c->dW, c->dWsnap, (size_t)c->mp * c->KP * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->w_implicit = false;
  c->ps_valid = false; c->trace_ready = false; c->num_valid = false;
  if (c->dSing) HIPCHK(c, hipMemsetAsync(c->dSing, 0, sizeof(int), c->stream));
  return PMF_OK;
}

// H with the state derived from it: for NMF / BNMF the Gram matrix G = H H^T as the last H step left it (whole, or as the
// per-workgroup partial sums the next one-pass launch adds) -- a restored H then continues with the SAME bits of G; the other
// classes form G from H whenever they need it.
int pmf_snapshot_h(pmf_ctx* c) {
  PMFCHK(need(c, false, false, true));
  const size_t hb = (size_t)c->KP * c->np * sizeof(float), gb = (size_t)c->KP * c->KP * sizeof(float);
  if (!c->dHsnap) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->dHsnap), hb + gb + (size_t)PMF_HGRAM_MAX_WGS * gb));
  HIPCHK(c, hipMemcpyAsync(c->dHsnap, c->dH, hb, hipMemcpyDeviceToDevice, c->stream));
  if (c->dHd) {                    // SNMF: the float64 H with it (a restored H continues with the same bits)
    if (!c->dHdSnap) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->dHdSnap), 2 * hb));
    HIPCHK(c, hipMemcpyAsync(c->dHdSnap, c->dHd, 2 * hb, hipMemcpyDeviceToDevice, c->stream));
  }
  const bool keep_g = (c->algo == PMF_ALGO_NMF || c->algo == PMF_ALGO_BNMF) && c->g_valid && c->dG != nullptr;
  c->hsnap_g_valid = keep_g;
  c->hsnap_g_parts = keep_g ? c->g_parts : 0;
  if (keep_g) {
    char* gs = reinterpret_cast<char*>(c->dHsnap) + hb;
    HIPCHK(c, hipMemcpyAsync(gs, c->dG, gb, hipMemcpyDeviceToDevice, c->stream));
    if (c->g_parts > 0 && c->dGpart)
      HIPCHK(c, hipMemcpyAsync(gs + gb, c->dGpart, (size_t)c->g_parts * gb, hipMemcpyDeviceToDevice, c->stream));
  }
  c->hsnap_valid = true;
  return PMF_OK;
}

int pmf_restore_h(pmf_ctx* c) {
  if (!c) return PMF_EINVAL;
  if (!c->hsnap_valid) return fail(c, PMF_EINVAL, "pmf_restore_h: no snapshot (pmf_snapshot_h)");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t hb = (size_t)c->KP * c->np * sizeof(float), gb = (size_t)c->KP * c->KP * sizeof(float);
  HIPCHK(c, hipMemcpyAsync(c->dH, c->dHsnap, hb, hipMemcpyDeviceToDevice, c->stream));
  if (c->dHd && c->dHdSnap) HIPCHK(c, hipMemcpyAsync(c->dHd, c->dHdSnap, 2 * hb, hipMemcpyDeviceToDevice, c->stream));
  c->hd_synced = false;
  c->have_h = true; c->num_valid = false; c->trace_ready = false;
  c->g_valid = false; c->g_parts = 0;
  if (c->hsnap_g_valid) {
    const char* gs = reinterpret_cast<const char*>(c->dHsnap) + hb;
    HIPCHK(c, hipMemcpyAsync(c->dG, gs, gb, hipMemcpyDeviceToDevice, c->stream));
    if (c->hsnap_g_parts > 0)
      HIPCHK(c, hipMemcpyAsync(c->dGpart, gs + gb, (size_t)c->hsnap_g_parts * gb, hipMemcpyDeviceToDevice, c->stream));
    c->g_valid = true; c->g_parts = c->hsnap_g_parts;
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PMF_OK;
}

int pmf_kernel_exec_flops(pmf_ctx* c, double* executed_flops_per_launch) {
  if (!c || !executed_flops_per_launch) return PMF_EINVAL;
  *executed_flops_per_launch = c->stat.exec_flops;
  return PMF_OK;
}

int pmf_kernel_launch_ms(pmf_ctx* c, double* out_ms, int64_t cap, int64_t* count) {
  if (!c || (cap > 0 && !out_ms)) return PMF_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const int64_t pairs = (int64_t)(c->stat.used / 2);
  for (int64_t q = 0; q < pairs && q < cap; ++q) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->stat.ev[2 * q], c->stat.ev[2 * q + 1]));
    out_ms[q] = ms;
  }
  if (count) *count = pairs;
  return PMF_OK;
}

int pmf_abort(pmf_ctx* c, int32_t on) {
  if (!c) return PMF_EINVAL;
  c->abort_flag.store(on ? 1 : 0, std::memory_order_relaxed);
  return PMF_OK;
}

int pmf_synchronize(pmf_ctx* c) {
  if (!c) return PMF_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return PMF_OK;
}

}  // extern "C"
