"""ctypes binding of libpymf_hip.so (C ABI in include/pymf_hip.h).

The shared library is built in-tree by pymf_amd/csrc/build.py (hipcc,
--offload-arch=gfx950).  There is no CPU fallback: if the library is missing or
cannot be loaded, every entry point raises.
"""
import ctypes
import os

import numpy as np

# Single-node design (ranks of one xGMI node): keep RCCL's bootstrap off InfiniBand probing, which
# can stall communicator creation for minutes on hosts without a fabric; users may override both.
os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PMF_LIB") or os.path.join(_HERE, "csrc", "libpymf_hip.so")   # PMF_LIB: A/B builds

PMF_OK, PMF_EINVAL, PMF_EHIP, PMF_ENCCL, PMF_ENOMEM, PMF_ESINGULAR = 0, -1, -2, -3, -4, -5
ALGO_NMF, ALGO_NMFALS, ALGO_SNMF, ALGO_BNMF, ALGO_RNMF = 0, 1, 2, 3, 4
COMPUTE_W, COMPUTE_H, COMPUTE_ERR = 1, 2, 4
STREAM_RESID = 8
NCCL_ID_BYTES = 128
IPC_HANDLE_BYTES = 64

# every symbol include/pymf_hip.h declares: (name, restype, argtypes)
_c = ctypes
_ctx = _c.c_void_p
_fp = _c.POINTER(_c.c_float)
SYMBOLS = [
    ("pmf_device_count", _c.c_int, [_c.POINTER(_c.c_int32)]),
    ("pmf_nccl_unique_id", _c.c_int, [_c.c_void_p]),
    ("pmf_ctx_create", _c.c_int, [_c.POINTER(_ctx), _c.c_int32, _c.c_int64, _c.c_int64, _c.c_int32,
                                  _c.c_int32, _c.c_int32, _c.c_int32, _c.c_void_p]),
    ("pmf_ctx_destroy", _c.c_int, [_ctx]),
    ("pmf_last_error", _c.c_char_p, [_ctx]),
    ("pmf_set_v_dense_f32", _c.c_int, [_ctx, _c.c_void_p, _c.c_int64]),
    ("pmf_set_v_dense_f64", _c.c_int, [_ctx, _c.c_void_p, _c.c_int64]),
    ("pmf_set_v_csr_f32", _c.c_int, [_ctx, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64]),
    ("pmf_fill_v_uniform", _c.c_int, [_ctx, _c.c_uint64, _c.c_int64]),
    ("pmf_set_w_f32", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_get_w_f32", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_set_h_f32", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_get_h_f32", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_set_w_f64", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_get_w_f64", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_set_h_f64", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_get_h_f64", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_fill_w_uniform", _c.c_int, [_ctx, _c.c_uint64, _c.c_int64]),
    ("pmf_fill_h_uniform", _c.c_int, [_ctx, _c.c_uint64]),
    ("pmf_update_w", _c.c_int, [_ctx]),
    ("pmf_update_h", _c.c_int, [_ctx]),
    ("pmf_frobenius", _c.c_int, [_ctx, _c.POINTER(_c.c_double)]),
    ("pmf_factorize", _c.c_int, [_ctx, _c.c_int32, _c.c_uint32, _c.c_double, _c.c_void_p,
                                 _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32)]),
    ("pmf_set_lambda", _c.c_int, [_ctx, _c.c_double, _c.c_double]),
    ("pmf_get_lambda", _c.c_int, [_ctx, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
    ("pmf_rnmf_update_s", _c.c_int, [_ctx]),
    ("pmf_rnmf_get_s_f32", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_rnmf_set_s_f32", _c.c_int, [_ctx, _c.c_void_p]),
    ("pmf_nndsvd_init", _c.c_int, [_ctx, _c.POINTER(_c.c_int32)]),
    ("pmf_stream_begin", _c.c_int, [_ctx, _c.c_uint32, _c.c_int64]),
    ("pmf_stream_tile", _c.c_int, [_ctx, _c.c_int64, _c.c_int64, _c.c_void_p, _c.c_int64]),
    ("pmf_stream_end", _c.c_int, [_ctx, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int32)]),
    ("pmf_last_loop_ms", _c.c_int, [_ctx, _c.POINTER(_c.c_double)]),
    ("pmf_profile_enable", _c.c_int, [_ctx, _c.c_int32]),
    ("pmf_kernel_stats", _c.c_int, [_ctx, _c.POINTER(_c.c_char_p), _c.POINTER(_c.c_int64),
                                    _c.POINTER(_c.c_double), _c.POINTER(_c.c_double),
                                    _c.POINTER(_c.c_double)]),
    ("pmf_host_checksum", _c.c_int, [_c.c_void_p, _c.c_uint64, _c.POINTER(_c.c_uint64)]),
    ("pmf_set_option", _c.c_int, [_ctx, _c.c_char_p, _c.c_int64]),
    ("pmf_set_host_allreduce", _c.c_int, [_ctx, _c.c_void_p, _c.c_void_p]),
    ("pmf_ipc_export", _c.c_int, [_ctx, _c.c_int32, _c.c_int32, _c.c_void_p]),
    ("pmf_ipc_import", _c.c_int, [_ctx, _c.c_void_p, _c.c_int32]),
    ("pmf_ipc_selftest", _c.c_int, [_ctx, _c.c_int32, _c.POINTER(_c.c_int32)]),
    ("pmf_collective_name", _c.c_char_p, [_ctx]),
    ("pmf_invalidate_v", _c.c_int, [_ctx]),
    ("pmf_snapshot_w", _c.c_int, [_ctx]),
    ("pmf_restore_w", _c.c_int, [_ctx]),
    ("pmf_snapshot_h", _c.c_int, [_ctx]),
    ("pmf_restore_h", _c.c_int, [_ctx]),
    ("pmf_collective_ms", _c.c_int, [_ctx, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64)]),
    ("pmf_kernel_exec_flops", _c.c_int, [_ctx, _c.POINTER(_c.c_double)]),
    ("pmf_kernel_launch_ms", _c.c_int, [_ctx, _c.c_void_p, _c.c_int64, _c.POINTER(_c.c_int64)]),
    ("pmf_nnqp_counters", _c.c_int, [_ctx, _c.POINTER(_c.c_int64), _c.c_int32]),
    ("pmf_synchronize", _c.c_int, [_ctx]),
    ("pmf_abort", _c.c_int, [_ctx, _c.c_int32]),
    ("pmf_path_name", _c.c_char_p, [_ctx]),
]

HOST_ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32)

_lib = None


class PmfError(RuntimeError):
    """A negative status from libpymf_hip; `.code` is the PMF_E* value (include/pymf_hip.h)."""
    code = None


def load():
    """dlopen libpymf_hip.so and bind every declared symbol.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PmfError("libpymf_hip.so not built: run `python -m pymf_amd.csrc.build` "
                       "(hipcc, gfx950). There is no CPU fallback.")
    if not os.environ.get("PMF_LIB"):                 # (an A/B build named explicitly is the caller's business)
        from .csrc import build as _build
        if _build.built_hash() != _build.source_hash():
            raise PmfError("libpymf_hip.so was not built from the sources at hand (hash of csrc/*.h, csrc/*.hip, "
                           "include/pymf_hip.h and the flags differs from libpymf_hip.so.srchash): run "
                           "`python -m pymf_amd.csrc.build`. A stale binary is never loaded.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, ctx=None):
    if rc == PMF_ESINGULAR:                 # what the reference's np.linalg.inv raises (snmf.py:69)
        raise np.linalg.LinAlgError("Singular matrix")
    if rc != PMF_OK:
        msg = load().pmf_last_error(ctx)
        err = PmfError("libpymf_hip error %d: %s" % (rc, (msg or b"").decode("utf-8", "replace")))
        err.code = int(rc)
        raise err


def device_count():
    n = ctypes.c_int32(0)
    rc = load().pmf_device_count(ctypes.byref(n))
    if rc != PMF_OK:
        return 0
    return int(n.value)


def nccl_unique_id():
    buf = ctypes.create_string_buffer(NCCL_ID_BYTES)
    check(load().pmf_nccl_unique_id(buf))
    return buf.raw


def host_checksum(a):
    """(shape, dtype, 128-bit digest of the bytes) of a host array: the change detector of the host
    classes (pmf_host_checksum; runs without a GPU).  Non-contiguous input is digested through a
    contiguous copy."""
    a = np.asarray(a)
    c = a if a.flags.c_contiguous else np.ascontiguousarray(a)
    out = (ctypes.c_uint64 * 2)()
    check(load().pmf_host_checksum(c.ctypes.data if c.size else None, c.nbytes, out))
    return (a.shape, a.dtype.str, int(out[0]), int(out[1]))


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Context(object):
    """Thin owner of one pmf_ctx*."""

    def __init__(self, algo, m_local, n, k, device=0, rank=0, nranks=1, nccl_id=None):
        self._lib = load()
        self._h = ctypes.c_void_p()
        idbuf = ctypes.create_string_buffer(nccl_id, NCCL_ID_BYTES) if nccl_id else None
        check(self._lib.pmf_ctx_create(ctypes.byref(self._h), algo, m_local, n, k, device, rank,
                                       nranks, idbuf))
        self.m, self.n, self.k = int(m_local), int(n), int(k)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.pmf_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        check(rc, self._h)

    @property
    def path_name(self):
        return self._lib.pmf_path_name(self._h).decode()

    def set_v_dense(self, V):
        V = np.asarray(V)
        assert V.shape == (self.m, self.n)
        if V.dtype == np.float64 and V.flags.c_contiguous:       # rounded on the device: no host conversion pass
            self._chk(self._lib.pmf_set_v_dense_f64(self._h, V.ctypes.data, V.shape[1]))
            return
        if V.dtype != np.float32 or not V.flags.c_contiguous:
            V = _f32c(V)
        self._chk(self._lib.pmf_set_v_dense_f32(self._h, V.ctypes.data, V.shape[1]))

    def set_v_csr(self, indptr, indices, vals):
        indptr = np.ascontiguousarray(indptr, dtype=np.int64)
        indices = np.ascontiguousarray(indices, dtype=np.int32)
        vals = _f32c(vals)
        assert indptr.shape[0] == self.m + 1
        self._chk(self._lib.pmf_set_v_csr_f32(self._h, indptr.ctypes.data, indices.ctypes.data,
                                              vals.ctypes.data, int(vals.shape[0])))

    def fill_v_uniform(self, seed, row0=0):
        self._chk(self._lib.pmf_fill_v_uniform(self._h, seed, row0))

    def fill_w_uniform(self, seed, row0=0):
        self._chk(self._lib.pmf_fill_w_uniform(self._h, seed, row0))

    def fill_h_uniform(self, seed):
        self._chk(self._lib.pmf_fill_h_uniform(self._h, seed))

    def _set_factor(self, A, shape, f32_fn, f64_fn):
        A = np.asarray(A)
        assert A.shape == shape, (A.shape, shape)
        if A.dtype == np.float64 and A.flags.c_contiguous:       # the reference's default dtype: rounded on the device
            self._chk(f64_fn(self._h, A.ctypes.data))
        else:
            A = _f32c(A)
            self._chk(f32_fn(self._h, A.ctypes.data))

    def set_w(self, W):
        self._set_factor(W, (self.m, self.k), self._lib.pmf_set_w_f32, self._lib.pmf_set_w_f64)

    def set_h(self, H):
        self._set_factor(H, (self.k, self.n), self._lib.pmf_set_h_f32, self._lib.pmf_set_h_f64)

    def get_w(self):
        W = np.empty((self.m, self.k), dtype=np.float32)
        self._chk(self._lib.pmf_get_w_f32(self._h, W.ctypes.data))
        return W

    def get_h(self):
        H = np.empty((self.k, self.n), dtype=np.float32)
        self._chk(self._lib.pmf_get_h_f32(self._h, H.ctypes.data))
        return H

    def _get_into(self, out, shape, f32_fn, f64_fn):
        """Write a factor straight into the caller's array when that is a C-contiguous float32 / float64 array of the
        right shape (float64: widened on the device); returns False when it is not (the caller copies)."""
        if type(out) is not np.ndarray or out.shape != shape or not out.flags.c_contiguous or not out.flags.writeable:
            return False
        if out.dtype == np.float64:
            self._chk(f64_fn(self._h, out.ctypes.data))
        elif out.dtype == np.float32:
            self._chk(f32_fn(self._h, out.ctypes.data))
        else:
            return False
        return True

    def get_w_into(self, out):
        return self._get_into(out, (self.m, self.k), self._lib.pmf_get_w_f32, self._lib.pmf_get_w_f64)

    def get_h_into(self, out):
        return self._get_into(out, (self.k, self.n), self._lib.pmf_get_h_f32, self._lib.pmf_get_h_f64)

    def update_w(self):
        self._chk(self._lib.pmf_update_w(self._h))

    def update_h(self):
        self._chk(self._lib.pmf_update_h(self._h))

    def frobenius(self):
        out = ctypes.c_double(0.0)
        self._chk(self._lib.pmf_frobenius(self._h, ctypes.byref(out)))
        return float(out.value)

    def factorize(self, niter, compute_w=True, compute_h=True, compute_err=True, conv_eps=1e-8):
        """Returns (ferr ndarray or None, iters_done, converged_at)."""
        flags = (COMPUTE_W if compute_w else 0) | (COMPUTE_H if compute_h else 0) | \
                (COMPUTE_ERR if compute_err else 0)
        ferr = np.zeros(max(int(niter), 1), dtype=np.float64) if compute_err else None
        done = ctypes.c_int32(0)
        conv = ctypes.c_int32(-1)
        self._chk(self._lib.pmf_factorize(self._h, int(niter), flags, float(conv_eps),
                                          ferr.ctypes.data if compute_err else None,
                                          ctypes.byref(done), ctypes.byref(conv)))
        if compute_err:
            ferr = ferr[:int(niter)]
        return ferr, int(done.value), int(conv.value)

    def set_lambda(self, lamb_w, lamb_h):
        self._chk(self._lib.pmf_set_lambda(self._h, float(lamb_w), float(lamb_h)))

    def get_lambda(self):
        a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._chk(self._lib.pmf_get_lambda(self._h, ctypes.byref(a), ctypes.byref(b)))
        return float(a.value), float(b.value)

    def rnmf_update_s(self):
        self._chk(self._lib.pmf_rnmf_update_s(self._h))

    def rnmf_set_s(self, S):
        S = _f32c(S)
        assert S.shape == (self.m, self.n), (S.shape, self.m, self.n)
        self._chk(self._lib.pmf_rnmf_set_s_f32(self._h, S.ctypes.data))

    def rnmf_get_s(self):
        S = np.empty((self.m, self.n), dtype=np.float32)
        self._chk(self._lib.pmf_rnmf_get_s_f32(self._h, S.ctypes.data))
        return S

    def stream_begin(self, compute_w=True, compute_h=True, compute_err=True, max_tile_rows=65536, resid=False):
        """Open one streamed pass over V (pmf_stream_begin); resid=True: direct residual pass only."""
        flags = STREAM_RESID if resid else \
            ((COMPUTE_W if compute_w else 0) | (COMPUTE_H if compute_h else 0) | (COMPUTE_ERR if compute_err else 0))
        self._chk(self._lib.pmf_stream_begin(self._h, flags, int(max_tile_rows)))
        self._tiles_alive = []

    def stream_tile(self, row0, tile):
        """Hand rows [row0, row0 + len(tile)) to the pass; `tile` is kept alive until two tiles later."""
        t = np.ascontiguousarray(tile, dtype=np.float32)
        if t.ndim != 2 or t.shape[1] != self.n:
            raise ValueError("tile must be rows x %d" % self.n)
        self._chk(self._lib.pmf_stream_tile(self._h, int(row0), t.shape[0], t.ctypes.data, t.shape[1]))
        self._tiles_alive = (self._tiles_alive + [t])[-3:]

    def stream_end(self):
        """Close the pass: returns (ferr or None, needs_direct)."""
        ferr = ctypes.c_double(-1.0)
        nd = ctypes.c_int32(0)
        self._chk(self._lib.pmf_stream_end(self._h, ctypes.byref(ferr), ctypes.byref(nd)))
        self._tiles_alive = []
        return float(ferr.value), bool(nd.value)

    def nndsvd_init(self):
        """W, H <- NNDSVD of the resident V (pymf/nndsvd.py:79-106); returns the rank found."""
        found = ctypes.c_int32(0)
        self._chk(self._lib.pmf_nndsvd_init(self._h, ctypes.byref(found)))
        return int(found.value)

    def last_loop_ms(self):
        ms = ctypes.c_double(0.0)
        self._chk(self._lib.pmf_last_loop_ms(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def profile_enable(self, on=True):
        self._chk(self._lib.pmf_profile_enable(self._h, 1 if on else 0))

    def kernel_stats(self):
        name = ctypes.c_char_p()
        n = ctypes.c_int64(0)
        ms = ctypes.c_double(0.0)
        fl = ctypes.c_double(0.0)
        by = ctypes.c_double(0.0)
        self._chk(self._lib.pmf_kernel_stats(self._h, ctypes.byref(name), ctypes.byref(n),
                                             ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)))
        ex = ctypes.c_double(0.0)
        self._chk(self._lib.pmf_kernel_exec_flops(self._h, ctypes.byref(ex)))
        return dict(name=(name.value or b"").decode(), launches=int(n.value), mean_ms=float(ms.value),
                    flops_per_launch=float(fl.value), bytes_per_launch=float(by.value),
                    executed_flops_per_launch=float(ex.value))

    def set_option(self, name, value):
        self._chk(self._lib.pmf_set_option(self._h, name.encode(), int(value)))

    def set_host_allreduce(self, reduce_array):
        """Route the cross-rank sums through `reduce_array(ndarray) -> ndarray` (the sum over all ranks)
        instead of RCCL: pmf_set_host_allreduce.  For plumbing checks with ranks that share one GPU."""
        def _cb(user, buf, count, is_f64):
            try:
                dt = np.float64 if is_f64 else np.float32
                a = np.ctypeslib.as_array(ctypes.cast(buf, ctypes.POINTER(ctypes.c_double if is_f64 else ctypes.c_float)),
                                          shape=(int(count),))
                a[:] = np.asarray(reduce_array(a.copy()), dtype=dt)
                return 0
            except Exception:            # never let an exception cross the C boundary
                return 1
        self._host_ar_cb = HOST_ALLREDUCE_FN(_cb)      # keep the trampoline alive
        self._chk(self._lib.pmf_set_host_allreduce(self._h, ctypes.cast(self._host_ar_cb, ctypes.c_void_p), None))

    def ipc_export(self, rank, nranks):
        """This rank's IPC handle of its receive area (pmf_ipc_export), IPC_HANDLE_BYTES bytes."""
        buf = ctypes.create_string_buffer(IPC_HANDLE_BYTES)
        self._chk(self._lib.pmf_ipc_export(self._h, int(rank), int(nranks), buf))
        return buf.raw

    def ipc_import(self, handles):
        """Map the peers' receive areas: `handles` = every rank's handle in rank order (pmf_ipc_import)."""
        assert all(len(p_) == IPC_HANDLE_BYTES for p_ in handles)
        allh = ctypes.create_string_buffer(b"".join(handles), IPC_HANDLE_BYTES * len(handles))
        self._chk(self._lib.pmf_ipc_import(self._h, allh, len(handles)))

    def ipc_selftest(self, rounds=6):
        """True iff the one-shot all-reduce reproduced the other transport's sums `rounds` times (pmf_ipc_selftest)."""
        ok = ctypes.c_int32(0)
        self._chk(self._lib.pmf_ipc_selftest(self._h, int(rounds), ctypes.byref(ok)))
        return bool(ok.value)

    @property
    def collective_name(self):
        return (self._lib.pmf_collective_name(self._h) or b"").decode()

    def invalidate_v(self):
        self._chk(self._lib.pmf_invalidate_v(self._h))

    def snapshot_w(self):
        """Device-side copy of W before a step that may fail (pmf_snapshot_w)."""
        self._chk(self._lib.pmf_snapshot_w(self._h))

    def restore_w(self):
        self._chk(self._lib.pmf_restore_w(self._h))

    def snapshot_h(self):
        """Device-side copy of H and of the Gram state derived from it (pmf_snapshot_h)."""
        self._chk(self._lib.pmf_snapshot_h(self._h))

    def restore_h(self):
        self._chk(self._lib.pmf_restore_h(self._h))

    def kernel_launch_ms(self, cap=65536):
        """Durations (ms) of the dominant kernel's launches since profile_enable(), in launch order."""
        out = np.zeros(int(cap), dtype=np.float64)
        n = ctypes.c_int64(0)
        self._chk(self._lib.pmf_kernel_launch_ms(self._h, out.ctypes.data, int(cap), ctypes.byref(n)))
        return out[:min(int(n.value), int(cap))]

    def nnqp_counters(self, reset=False):
        """Live counts of k_nnqp_quad over the W half steps since the last reset (pmf_nnqp_counters)."""
        out = (ctypes.c_int64 * 8)()
        self._chk(self._lib.pmf_nnqp_counters(self._h, out, 1 if reset else 0))
        keys = ("wave_tasks", "passes", "sum_largest_system", "problems")
        return {"frame16": dict(zip(keys, (int(x) for x in out[0:4]))), "frame32": dict(zip(keys, (int(x) for x in out[4:8])))}

    def collective_ms(self):
        """(mean ms, count) of the per-iteration collective's launches since profile_enable() (pmf_collective_ms)."""
        ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
        self._chk(self._lib.pmf_collective_ms(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def synchronize(self):
        self._chk(self._lib.pmf_synchronize(self._h))

    def abort(self, on=True):
        """Ask the running (or next) factorize() of this context to return early / clear the request (pmf_abort; the one
        call that may come from a second thread)."""
        self._chk(self._lib.pmf_abort(self._h, 1 if on else 0))
