/*
 * pymf_hip.h -- C ABI of libpymf_hip.so: the MI355X (gfx950) implementation of the
 * iterative factorize() hot path of nils-werner/pymf (NMF / NMFALS / SNMF).
 *
 * Plain C, no C++ or torch types.  One pmf_ctx drives ONE GPU from one host
 * thread (thread-compatible, not thread-safe); multi-GPU = one process (ctx)
 * per GPU, rows of V/W sharded, ONE cross-rank sum of (W^T V | W^T W) per iteration
 * inside pmf_factorize (for the single hooks: inside whichever of pmf_update_w /
 * pmf_update_h forms those sums -- all ranks make the same calls in the same order).
 * How that sum travels: for 2..8 ranks of one node, once
 * pmf_ipc_export / pmf_ipc_import have mapped the peers' receive areas and pmf_ipc_selftest
 * has agreed with the other transport on every rank, a ONE-SHOT exchange -- every rank writes
 * its partial into every peer's memory and adds the N partials in rank order -- folded into
 * the slab-reduce and H-step launches of the loop (no launch of its own); otherwise, and for
 * payloads beyond 256 KiB, ncclAllReduce on the context's RCCL communicator (or the host
 * callback of pmf_set_host_allreduce where the ranks cannot form one).  Every transport adds
 * in rank order: H stays bit-identical across the ranks.  The caller owns every host buffer;
 * the library owns all device memory, its HIP stream and its RCCL communicator.
 *
 * Reference interface each entry point replaces (paths into nils-werner/pymf):
 *   pmf_ctx_create        NMF.__init__                 pymf/nmf.py:71-97
 *   pmf_set_v_*           self.data (aliased ndarray)  pymf/nmf.py:93,123,129 (data[:,:])
 *   pmf_set_w/h, get_w/h  self.W / self.H attributes   pymf/nmf.py:116-120,173-177
 *   pmf_update_w          NMF.update_w                 pymf/nmf.py:128-132
 *                         SNMF.update_w                pymf/snmf.py:67-70
 *                         NMFALS.update_w              pymf/nmfals.py:85-97
 *   pmf_update_h          NMF.update_h                 pymf/nmf.py:122-126
 *                         SNMF.update_h                pymf/snmf.py:72-91
 *                         NMFALS.update_h              pymf/nmfals.py:70-82
 *                         BNMF.update_w / update_h     pymf/bnmf.py:79-90 (algo 3, 'next' row)
 *   pmf_frobenius         NMF.frobenius_norm           pymf/nmf.py:100-114
 *   pmf_factorize         NMF.factorize loop body      pymf/nmf.py:182-202
 *                         (incl. NMF.converged         pymf/nmf.py:134-139)
 *   pmf_rnmf_update_s     RNMF.update_s                pymf/rnmf.py:96-98   (algo 4, 'next' row)
 *   pmf_nndsvd_init       NNDSVD.update_w + SVD        pymf/nndsvd.py:78-106, pymf/svd.py:105-148
 *   pmf_stream_*          the data[:,:] reads of       pymf/nmf.py:123,129 for data that is not resident
 *
 * Every function returns PMF_OK (0) or a negative status and never throws;
 * pmf_last_error() gives a human-readable message for the last failure.
 */
#ifndef PYMF_HIP_H
#define PYMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmf_ctx pmf_ctx;

enum {
  PMF_OK = 0,
  PMF_EINVAL = -1,   /* bad argument / unsupported shape / call order */
  PMF_EHIP = -2,     /* a HIP runtime call failed */
  PMF_ENCCL = -3,    /* an RCCL call failed */
  PMF_ENOMEM = -4,   /* device or host allocation failed */
  PMF_ESINGULAR = -5 /* SNMF: H H^T is singular (the reference's np.linalg.inv raises LinAlgError, snmf.py:69) */
};

enum { PMF_ALGO_NMF = 0, PMF_ALGO_NMFALS = 1, PMF_ALGO_SNMF = 2, PMF_ALGO_BNMF = 3, PMF_ALGO_RNMF = 4 };

/* pmf_factorize flags (the reference's factorize() keyword arguments, nmf.py:141-142) */
enum { PMF_COMPUTE_W = 1u, PMF_COMPUTE_H = 2u, PMF_COMPUTE_ERR = 4u };

#define PMF_NCCL_ID_BYTES 128

/* Number of visible HIP devices. */
int pmf_device_count(int32_t* out);

/* Fill out[PMF_NCCL_ID_BYTES] with a fresh RCCL unique id (call on rank 0, hand the
 * bytes to every rank by any means, then pass them to pmf_ctx_create). */
int pmf_nccl_unique_id(void* out);

/*
 * Create a context on HIP device `device`.
 *   algo      PMF_ALGO_*
 *   m_local   rows of V (= rows of W) held by THIS rank (the whole matrix when nranks==1)
 *   n         columns of V (= columns of H), identical on all ranks
 *   k         num_bases, 1 .. 2432 (NMFALS / NMFNNLS: 1 .. 1024); beyond 128 -- 64 for NMFALS -- generic kernels run.
 *             (The reference has no limit, nmf.py:116-120; beyond these PMF_EINVAL.)
 *   rank,nranks,nccl_id   RCCL world; nranks==1 -> nccl_id may be NULL and RCCL is not touched
 *                         (a non-NULL id with nranks==1 creates a 1-rank communicator)
 *                         With nranks > 1 the ranks must agree on which cached sums are current -- they decide which
 *                         collectives the next call runs: when ANY rank has uploaded new V, W or H (pmf_set_*), EVERY rank
 *                         calls pmf_invalidate_v before the next update / factorize / frobenius call (the host classes
 *                         exchange one flag per call for this, pymf_amd/nmf.py _sync_to_device).
 */
int pmf_ctx_create(pmf_ctx** out, int32_t algo, int64_t m_local, int64_t n, int32_t k,
                   int32_t device, int32_t rank, int32_t nranks, const void* nccl_id);
int pmf_ctx_destroy(pmf_ctx* ctx);
const char* pmf_last_error(const pmf_ctx* ctx);   /* ctx may be NULL: last create error */

/* V: host row-major m_local x n float32 with leading dimension ld (elements). Copied. */
int pmf_set_v_dense_f32(pmf_ctx* ctx, const float* V, int64_t ld);
/* V as CSR (m_local rows): indptr[m_local+1] int64, indices[nnz] int32, vals[nnz] f32. */
int pmf_set_v_csr_f32(pmf_ctx* ctx, const int64_t* indptr, const int32_t* indices,
                      const float* vals, int64_t nnz);
/* Fill V on the device with the bench's synthetic U[0,1) stream (counter-based, so
 * the values depend only on (seed, global row, column)); row0 = first global row. */
int pmf_fill_v_uniform(pmf_ctx* ctx, uint64_t seed, int64_t row0);

/* The same for a float64 host array (the reference's default dtype, nmf.py:117,120): the bytes go up as they are and are
 * rounded to the device's float32 ON the device (no host-side conversion pass). */
int pmf_set_v_dense_f64(pmf_ctx* ctx, const double* V, int64_t ld);

/* W: m_local x k row-major (ld = k).  H: k x n row-major (ld = n). Host buffers. */
int pmf_set_w_f32(pmf_ctx* ctx, const float* W);
int pmf_get_w_f32(pmf_ctx* ctx, float* W);
int pmf_set_h_f32(pmf_ctx* ctx, const float* H);
int pmf_get_h_f32(pmf_ctx* ctx, float* H);
/* ... and for float64 host arrays -- what self.W / self.H are by default (nmf.py:117,120): rounded to / widened from the
 * device's float32 on the device, so that neither direction needs a conversion pass over m x k on the host.
 * SNMF (option "snmf_h64", default on): H is KEPT in float64 on the device -- pmf_set_h_f64 / pmf_get_h_f64 carry
 * the float64 values as they are; pmf_get_h_f32 returns their rounding. */
int pmf_set_w_f64(pmf_ctx* ctx, const double* W);
int pmf_get_w_f64(pmf_ctx* ctx, double* W);
int pmf_set_h_f64(pmf_ctx* ctx, const double* H);
int pmf_get_h_f64(pmf_ctx* ctx, double* H);
int pmf_fill_w_uniform(pmf_ctx* ctx, uint64_t seed, int64_t row0);
int pmf_fill_h_uniform(pmf_ctx* ctx, uint64_t seed);

/* One hook each (blocking until done on the device).  On shapes the one-pass kernel takes,
 * pmf_update_w also leaves (W^T V | W^T W) of the new W cached for the pmf_update_h that follows. */
int pmf_update_w(pmf_ctx* ctx);
int pmf_update_h(pmf_ctx* ctx);
int pmf_frobenius(pmf_ctx* ctx, double* out);   /* sqrt(sum((V - W H)^2)), all ranks' rows */

/*
 * The factorize() loop: for i in [0,niter): [update_w] [update_h] [ferr[i] = frobenius]
 * and, when PMF_COMPUTE_ERR and i > 1, stop if |ferr[i]-ferr[i-1]|/n < conv_eps
 * (nmf.py:134-139,198-202).  On convergence at iteration i, *converged_at = i
 * (the caller truncates ferr to ferr[:i] as the reference does) else -1.
 * *iters_done = number of loop bodies executed.  ferr may be NULL without PMF_COMPUTE_ERR.
 * Blocks until the device loop has finished.
 */
int pmf_factorize(pmf_ctx* ctx, int32_t niter, uint32_t flags, double conv_eps,
                  double* ferr, int32_t* iters_done, int32_t* converged_at);

/* BNMF only (pymf/bnmf.py:79-90,118-119): the penalty weights _lamb_W/_lamb_H used by the next
 * update_w/update_h; every update_h multiplies both by 1.1 (bnmf.py:84-85), as the reference does.
 * The caller sets 1/niter before each factorize() like BNMF.factorize. */
int pmf_set_lambda(pmf_ctx* ctx, double lamb_w, double lamb_h);
int pmf_get_lambda(pmf_ctx* ctx, double* lamb_w, double* lamb_h);

/* RNMF only (pymf/rnmf.py, algo 4; pmf_set_lambda(ctx, lamb, 0) sets the soft threshold _lamb):
 * pmf_rnmf_update_s = RNMF.update_s (rnmf.py:96-98), S = soft_thresholding(data - W H, lamb);
 * update_h runs it itself afterwards as the reference does (rnmf.py:107).  pmf_rnmf_get_s_f32
 * copies S (m_local x n, row-major) to the host. */
int pmf_rnmf_update_s(pmf_ctx* ctx);
int pmf_rnmf_get_s_f32(pmf_ctx* ctx, float* S);
/* S [m][n] contiguous handed to a context (the reference's S is an attribute that copies and pickles of the object carry,
 * rnmf.py:96-98; new data keeps it too: pmf_set_v_* on an RNMF context re-bases the device state D = S - data). */
int pmf_rnmf_set_s_f32(pmf_ctx* ctx, const float* S);

/* Streamed V (the `data[:, :]` idiom of nmf.py:123,129 for data that does not fit
 * in HBM: an h5py dataset, a memmap, a matrix beyond 288 GB).  NMF, BNMF, SNMF and NMFALS contexts (not RNMF: the reference's RNMF keeps S, an in-memory array of data's shape, rnmf.py:94-98); pmf_set_v_* is not
 * called.  One pass = one iteration of the reference loop (nmf.py:183-202):
 *   pmf_stream_begin(ctx, flags, max_tile_rows)   flags as pmf_factorize (PMF_COMPUTE_W/H/ERR)
 *   pmf_stream_tile(ctx, row0, rows, tile, ld)    row tiles in order; row0 and rows multiples of 64
 *                                                 (the last tile may be ragged); `tile` is row-major
 *                                                 host memory that must stay valid until the next but
 *                                                 one pmf_stream_tile call (or pmf_stream_end) has RETURNED:
 *                                                 the call waits for the copy of the tile two calls back
 *   pmf_stream_end(ctx, &ferr, &needs_direct)     H step; ferr = ||V - W H|| by the trace identity
 * needs_direct = 1 reports that the identity cancels (residual energy below 1e-3 of ||V||^2): a pass with
 * flags = PMF_STREAM_RESID evaluates sum((V - W H)^2) tile by tile instead (no update). */
#define PMF_STREAM_RESID 8u
int pmf_stream_begin(pmf_ctx* ctx, uint32_t flags, int64_t max_tile_rows);
int pmf_stream_tile(pmf_ctx* ctx, int64_t row0, int64_t rows, const float* tile, int64_t ld);
int pmf_stream_end(pmf_ctx* ctx, double* ferr, int32_t* needs_direct);

/* NNDSVD initialisation (pymf/nndsvd.py:79-108 = NNDSVD.update_w, with the SVD of pymf/svd.py:125-148):
 * fills the context's W and H from the dense V already set.  Needs n <= 16384 (the Gram matrix
 * data^T data is n x n: all its eigenpairs by a float64 Jacobi iteration up to 1024 columns, the num_bases largest
 * by a Chebyshev-filtered subspace iteration on the float64 MFMA beyond -- option "nndsvd_topk"; a wide matrix is handled by the caller on the transposed problem, as the
 * reference's SVD switches between its left and right forms, svd.py:237-246) and num_bases <= n.
 * rank_found (may be NULL) receives how many of the leading num_bases eigenvalues exceed the
 * reference's 1e-8 cut (svd.py:130-131); fewer than num_bases is PMF_EINVAL (the reference raises
 * IndexError).  Row-sharded contexts sum the Gram matrix and the split norms over all ranks. */
int pmf_nndsvd_init(pmf_ctx* ctx, int32_t* rank_found);

/* Device time (ms, HIP events on the library's stream) of the last pmf_factorize loop. */
int pmf_last_loop_ms(pmf_ctx* ctx, double* ms);

/*
 * Per-kernel timing for bench.py's roofline: when enabled, every launch of the
 * dominant kernel of the current algo is bracketed by HIP events on the
 * library's stream.  pmf_kernel_stats returns the kernel's name, launch count
 * and mean duration since the last reset.
 */
int pmf_profile_enable(pmf_ctx* ctx, int32_t on);
int pmf_kernel_stats(pmf_ctx* ctx, const char** name, int64_t* launches, double* mean_ms,
                     double* flops_per_launch, double* bytes_per_launch);

/* Host-only helper of the boundary (touches no device): an order-dependent 128-bit digest of `nbytes`
 * bytes of caller memory, out2[0..1].  The reference always computes from the CURRENT contents of
 * self.data / self.W / self.H (nmf.py:122-132); a host class that keeps device copies uses this to notice
 * in-place edits (permutations included) and re-upload.  Multi-threaded, memory speed. */
int pmf_host_checksum(const void* data, uint64_t nbytes, uint64_t* out2);

/* Tuning knobs (results agree to rounding whatever they say).  Known names:
 *   "oneshot_allreduce" 1 / 0: small cross-rank sums on the one-shot IPC all-reduce (after pmf_ipc_import) or on the
 *                context's other transport.
 *   "profile_every" N >= 1 (default 1): with pmf_profile_enable only every N-th launch of the dominant kernel (and of the
 *                per-iteration collective) carries HIP events -- a timed launch costs the loop about 5 us.
 *   "fold_exchange" 1 (default) / 0: inside pmf_factorize's one-pass NMF / BNMF loop the per-iteration sum of
 *                (W^T V | W^T W) rides on the launches around it -- the slab reduce pushes this rank's partial tiles into
 *                every peer's receive area, the H-step launch waits for the peers' flags in its prologue and adds the N
 *                partials in rank order -- instead of a k_ipc_allreduce launch of its own.  Bit-identical either way.
 *   "snmf_w_pipe" snmf_gram = 2 on CSR data: the W = V M write of iteration i runs on a stream of its own beside the k x n
 *                sized kernels of iteration i + 1 (they never read W; M is double buffered); the value is the number of
 *                workgroup slots the write launch leaves free so that those kernels can be placed while it runs (default
 *                32); 0: everything in stream order.  Bit-identical results.
 *   "nnqp_count" 1: the W half steps of NMFALS / NMFNNLS run the COUNTING instantiation of the sixteen-lanes-per-problem
 *                kernel (pmf_nnqp_counters; it costs the kernel 8 %); 0 (default): no counters in the loop.
 *   "snmf_gram"  SNMF loops with both updates on iterate in Gram space -- P = M^T (V^T V), S = P M on
 *                k x n sized data, W materialised once after the last iteration -- instead of one pass
 *                over V per iteration: -1 automatic (default: CSR data always, dense data from about
 *                n / 2k iterations on), 0 never, 1 whenever the shape allows (n <= 1024), 2 as 1 but W = V M is
 *                written in EVERY iteration (what the reference's update_w does; same results).
 *   "nnqp_quad"  NMFALS / NMFNNLS with num_bases <= 64 and a well-conditioned Hessian: the half step's QPs on the
 *                sixteen-lanes-per-problem kernel (block principal pivoting on the smaller of HA[P,P] / inv(HA)[N,N]):
 *                1 (default) from 16 384 problems per half step on, 2 always, 0 never (the lane-per-variable kernel).
 *                Same minimisers (they are unique).
 *   "nndsvd_topk" pmf_nndsvd_init's eigen-solver: -1 (default) full Jacobi up to 1024 columns and the top-k subspace
 *                iteration beyond, 1 / 0 force one of them where both apply (top-k needs num_bases + 16 <= n, Jacobi
 *                n <= 4096).  Same W, H to ~1e-8 (well inside the float32 accuracy of the Gram matrix).
 *   "colgemm_stream" 1 (default): the W^T V | W^T W partials of the two-pass path on k_colgemm_stream (V fragments straight
 *                into registers, requests interleaved with the MFMAs, W rows through LDS once per workgroup) where it
 *                applies (16 < num_bases <= 64; blocks of 128 bases); 0: k_colgemm.  Bit-identical results.
 *   "rowgemm_stream" 1 (default): plain products with a long contraction (V H^T of NMFALS / SNMF, W = V M^T) on
 *                k_rowgemm_stream (A fragments straight into registers, requests interleaved with the MFMAs);
 *                0: on k_rowgemm.  Bit-identical results (same order of summation).
 *   "nnqp_frame16" 1 (default): the sixteen-lanes-per-problem kernel first on a 16-slot frame (settled active sets factorise
 *                systems of about 8 unknowns: twelve waves around one LDS image of HA and inv(HA), 168 registers -- three
 *                waves per SIMD instead of two), problems that outgrow it on the 32-slot frame behind; 0: the 32-slot
 *                frame for all.  Bit-identical results.
 *   "nnqp_wave"  1 (default): NMFALS / NMFNNLS sub-problems at 64 < num_bases <= 128 on k_nnqp_wave (one wave per problem,
 *                block principal pivoting on the smaller of HA[P,P] / inv(HA)[N,N], LDL^T in registers); 0: k_nnqp_big (one
 *                variable at a time, the inverse image in global memory).  Same KKT point, float32 results equal to rounding.
 *   "snmf_h64"   1 (default): SNMF with num_bases <= 128 keeps H in FLOAT64 on the device (the reference's H is float64,
 *                pymf/nmf.py:120, and W = V H^T inv(H H^T) amplifies its rounding by sigma_max / sigma_min of H): G = H H^T,
 *                M^T = inv(G) H and the H step (snmf.py:72-91, on the float64 MFMA) read and write that copy; the float32 H is
 *                its rounding.  pmf_set_h_f64 / pmf_get_h_f64 round-trip the float64 values exactly; a float32 upload
 *                replaces them by the widened values.  0: H is float32 between the steps (rounds 1-5).
 *   "fuse_chain" NMFALS at 49..64 bases, 0 (default): the k x k chain of a half step as two launches (Gram / slab sum, then
 *                the inverse); bit 0 / bit 1: the W / H half step's chain as ONE launch whose last workgroup inverts the
 *                Hessian it has completed.  Bit-identical; measured 1-2 % slower per iteration (an A/B knob).
 *   "force_tiled" 1: every path of this context takes the any-shape two-pass kernels (k_rowgemm / k_colgemm)
 *                even where a one-pass kernel covers the shape; 0 gives the one-pass kernels back.  For tests
 *                and measurements of the any-shape kernels on the bench shapes. */
int pmf_set_option(pmf_ctx* ctx, const char* name, int64_t value);

/* Host transport for the cross-rank sums, for set-ups in which the ranks cannot form an RCCL communicator
 * (plumbing checks with several ranks sharing one GPU; ranks without a common fabric).  Contexts created
 * with nranks == 1 and no nccl_id only.  fn(user, buf, count, is_f64) must replace the `count` floats
 * (is_f64 = 0) or doubles (1) at host pointer `buf` by their sum over all ranks and return 0; every
 * rank's library calls it at the same points in the same order (where the RCCL build runs
 * ncclAllReduce).  Each call is a blocking device -> host -> device round trip: not a fast path. */
typedef int (*pmf_host_allreduce_fn)(void* user, void* buf, int64_t count, int32_t is_f64);
int pmf_set_host_allreduce(pmf_ctx* ctx, pmf_host_allreduce_fn fn, void* user);

/* One-shot all-reduce for the small per-iteration sums (the reduction over m of pymf/nmf.py:124-125 is what
 * the row sharding splits): every rank owns a receive area that all peers map through HIP IPC; ONE kernel per rank writes
 * its partial of (W^T V | W^T W) into every peer's area, raises a flag there, waits for the peers' flags and adds the N
 * partials in rank order -- bit-identical sums on all ranks, no ring / tree hops.  Payloads up to 256 KiB take it; larger
 * ones (the n x n float64 V^T V of the Gram-space SNMF loop, NNDSVD's Gram matrix) stay on RCCL / the host transport.
 *   pmf_ipc_export(ctx, rank, nranks, handle_out)   allocate + export this rank's area (sized for nranks; also the self-test's
 *                                                   buffers, so that nothing is allocated between the test's collectives)
 *   (hand every rank's handle to every rank by any means, in rank order)
 *   pmf_ipc_import(ctx, handles, nranks)            map the peers' areas (the SAME nranks; all or nothing: a failing
 *                                                   hipIpcOpenMemHandle closes what was opened); from here on small sums take
 *                                                   the one-shot path -- inside pmf_factorize's one-pass loop folded into the
 *                                                   slab-reduce and H-step launches (option "fold_exchange")
 * 2 <= nranks <= 8, one node (ranks on different GPUs of one xGMI hive, or -- for plumbing checks -- sharing a GPU).
 * pmf_collective_name: which transports this context's cross-rank sums use and how often each ran. */
#define PMF_IPC_HANDLE_BYTES 64
int pmf_ipc_export(pmf_ctx* ctx, int32_t rank, int32_t nranks, void* handle_out);
int pmf_ipc_import(pmf_ctx* ctx, const void* handles, int32_t nranks);
/* The one-shot path against the context's other transport (RCCL / host) on rank- and round-dependent payloads, `rounds`
 * times (both slots get reused; odd rounds in the split producer / consumer form the loop uses): *ok = 1 iff all rounds agreed
 * and no wait ran out (2 s per wait).  Every rank calls it at the same
 * point; the caller combines the verdicts and switches the path off on ALL ranks if any disagreed
 * (pmf_set_option(ctx, "oneshot_allreduce", 0)). */
int pmf_ipc_selftest(pmf_ctx* ctx, int32_t rounds, int32_t* ok);
const char* pmf_collective_name(pmf_ctx* ctx);

/* Forget everything derived from V (||V||^2, cached partial sums): for streamed `data` that the caller
 * rebound or edited between calls. */
int pmf_invalidate_v(pmf_ctx* ctx);

/* A W step that can fail -- SNMF: np.linalg.inv raises on a singular H H^T BEFORE W is rebound (snmf.py:69-70) --
 * must leave the previous W behind.  The host class keeps W on the device between calls, so before such a step it
 * asks for a device-side copy (pmf_snapshot_w: one device-to-device copy, W materialised first if it was implicit) and
 * puts it back when the step raised (pmf_restore_w).  No host traffic either way. */
int pmf_snapshot_w(pmf_ctx* ctx);
int pmf_restore_w(pmf_ctx* ctx);
/* The same for H (k x n: small), together with what the device derived from it (NMF / BNMF: the Gram matrix H H^T as the last
 * H step left it, so that a restored H continues with the same bits).  With pmf_snapshot_w / pmf_restore_w: the pair of factors
 * a loop started from, for a caller that finds out DURING the loop that it has to start again (pmf_abort). */
int pmf_snapshot_h(pmf_ctx* ctx);
int pmf_restore_h(pmf_ctx* ctx);

/* With pmf_profile_enable the per-iteration collective -- the sum of (W^T V | W^T W) over the ranks, what the row sharding of
 * pymf/nmf.py:124-125 costs -- is bracketed by HIP events as well: mean duration (ms) and count since the last enable. */
int pmf_collective_ms(pmf_ctx* ctx, double* mean_ms, int64_t* count);

/* The flops one launch of that kernel really executes (pmf_kernel_stats reports SURVEY's algorithmic
 * count): W^T W is symmetric (upper triangle only) and SNMF's W step is reassociated. */
int pmf_kernel_exec_flops(pmf_ctx* ctx, double* executed_flops_per_launch);

/* The individual launch durations behind pmf_kernel_stats' mean, in launch order: out_ms[0..min(cap,*count))
 * (ms, HIP events on the library's stream); *count = launches recorded since the last pmf_profile_enable. */
int pmf_kernel_launch_ms(pmf_ctx* ctx, double* out_ms, int64_t cap, int64_t* count);

/* NMFALS / NMFNNLS on the sixteen-lanes-per-problem kernel (pymf/nmfals.py:85-97, the row QPs of update_w): running totals
 * since the last reset, counted on the device by the kernel itself -- out8[0..3] for the 16-slot frame, out8[4..7] for the
 * 32-slot frame: wave tasks (4 problems each), passes (one solve per problem each), the sum over the passes of the largest
 * system among the wave's four problems (the size the frame-padded elimination runs over), problems solved on that frame.
 * bench.py's roofline block is built from them (live counts instead of recorded constants). */
int pmf_nnqp_counters(pmf_ctx* ctx, int64_t* out8, int32_t reset);

int pmf_synchronize(pmf_ctx* ctx);
/* The ONE entry point that may be called from a second host thread while pmf_factorize runs on the context: on = 1 makes the
 * running (or the next) pmf_factorize return at its next iteration / chunk boundary with *iters_done = what it completed;
 * on = 0 clears the request.  The host class runs its digest of `data` (the reference re-reads self.data[:,:] in every hook,
 * nmf.py:123,129) BESIDE the device loop and, should the bytes have changed since the upload, stops the loop, puts W / H back
 * (pmf_restore_w, pmf_restore_h), uploads and starts again. */
int pmf_abort(pmf_ctx* ctx, int32_t on);

/* Introspection used by tests: which code path update_w/update_h take for this shape.
 * Returns a static string such as "fused_k64_n256" or "tiled". */
const char* pmf_path_name(const pmf_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PYMF_HIP_H */
