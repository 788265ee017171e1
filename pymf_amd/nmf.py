"""pymf_amd.NMF -- drop-in for pymf.NMF (reference pymf/nmf.py) on MI355X.

Same class surface as the reference (`NMF(data, num_bases=4)`,
`.factorize(niter, show_progress, compute_w, compute_h, compute_err)`, `.W`,
`.H`, `.ferr`, `.frobenius_norm()`, the overridable hooks `init_w / init_h /
update_w / update_h / frobenius_norm / converged`, `_EPS`), but every update
runs as HIP kernels in libpymf_hip.so through a ctypes C ABI.
There is no NumPy fallback: without the library (or a GPU) the calls raise.

Template method (nmf.py:182-202): `factorize()` of an unmodified shipped class is
ONE C call (`pmf_factorize`).  When a subclass (or the instance) overrides any of
`update_w / update_h / frobenius_norm / converged`, or `show_progress=True` asks
for live log lines, `factorize()` runs the reference's loop hook by hook -- every
shipped hook is itself one C call -- so user plug-ins behave as in the reference.

Host-visible state follows the reference: W/H are created lazily, W first
(nmf.py:173-177), default float64 (nmf.py:117,120); NMF writes results back IN
PLACE into the existing arrays (nmf.py:125-126,131-132) keeping their dtype;
the arithmetic on the device is float32.  The device copies of data / W / H are
refreshed whenever the host arrays changed (an order-dependent digest of their
bytes, `pmf_host_checksum`), because the reference always computes from the
current host arrays.
"""
import logging
import os
import sys
import time
import warnings

import numpy as np

from . import _lib
from . import dist as _dist

__all__ = ["NMF"]


def _is_sparse(x):
    # (an object can only be a scipy.sparse matrix if somebody has imported scipy.sparse: importing it here costs a first
    # factorize() of dense data 0.1 s)
    sp = sys.modules.get("scipy.sparse")
    return sp is not None and sp.issparse(x)


def _fingerprint(a):
    """Change detector for a host array: shape, dtype and a position-sensitive 128-bit digest of
    its bytes (one multi-threaded pass at memory speed).  Any in-place edit -- a poke, a swap of two
    rows, a permutation of the bases -- changes it, so the device copy is refreshed."""
    return _lib.host_checksum(a)


def _draw_rows(m_total, ncols, lo, hi, chunk=65536):
    """rows [lo, hi) of np.random.random((m_total, ncols)) from the global legacy stream, drawn in
    chunks (a rank never holds the whole matrix) and leaving the stream exactly where the full draw
    would (so the H0 drawn next is the reference's, nmf.py:173-177)."""
    out = np.empty((hi - lo, ncols))
    r = 0
    while r < m_total:
        c = min(chunk, m_total - r)
        blk = np.random.random((c, ncols))
        a, b = max(r, lo), min(r + c, hi)
        if a < b:
            out[a - lo:b - lo] = blk[a - r:b - r]
        r += c
    return out


class PrecisionWarning(UserWarning):
    """float64 `data` is computed on in float32 (see NMF's class docstring)."""


class _LateDataCheck(object):
    """The digest of `data` on a second thread while pmf_factorize runs (NMF._late_data_check_ok).  Before the loop starts
    the device's W and H are copied device to device (pmf_snapshot_w / pmf_snapshot_h): what a restart puts back."""

    def __init__(self, owner, ctx, arr):
        import threading
        self.owner, self.ctx, self.arr = owner, ctx, arr
        self.fp, self.err = None, None
        ctx.snapshot_w()
        ctx.snapshot_h()
        self.thread = threading.Thread(target=self._run, name="pymf_amd-data-digest", daemon=True)
        self.thread.start()

    def _run(self):
        try:
            self.fp = _fingerprint(self.arr)                 # C, multi-threaded, the GIL released
            if self.fp != self.owner._v_fp:
                self.ctx.abort(True)                         # the loop is running on stale bytes: no point in finishing it
        except BaseException as e:                           # (never lose an error on the side thread)
            self.err = e

    def data_changed(self):
        self.thread.join()
        if self.err is not None:
            raise self.err
        return self.fp != self.owner._v_fp

    def finish(self):
        self.thread.join()
        self.ctx.abort(False)

    def restart(self):
        o, ctx = self.owner, self.ctx
        ctx.abort(False)
        ctx.restore_w()
        ctx.restore_h()                                      # (with the Gram matrix the last H step left: the same bits go on)
        o._warn_if_float64(self.arr)
        ctx.set_v_dense(self.arr)
        o._v_fp = self.fp


class NMF(object):
    """Non-negative matrix factorization, multiplicative updates (Lee & Seung).

    Precision: the reference's arithmetic follows its operands (float64 by default: nmf.py:117,120,122-132).
    Here every contraction runs on the float32 matrix cores and `data`, W and H are STORED in float32 on the
    device whatever the dtype of the host arrays (which keep their dtype: results are written back into
    float64 W / H arrays in place).  Results agree with the float64 reference within the tolerances of
    DESIGN.md section 4 (2e-5 relative Frobenius on W and H, 1e-5 on ferr after <= 50 iterations); float64
    `data` is rounded once on upload and a PrecisionWarning says so, once per object.
    `eager_factors = True` gives up the lazy write-back of W / H, which relies on CPython reference counts
    (other interpreters, tracers and debuggers that hold extra references are handled safely but eagerly).

    Parameters mirror the reference (pymf/nmf.py:23-66): data is m x n
    (m = _data_dimension rows, n = _num_samples columns), W is m x num_bases,
    H is num_bases x n.  Under a multi-rank world (one process per GPU, launched
    with RANK / WORLD_SIZE / MASTER_* set and `pymf_amd.dist.init_from_env()` called) `data`
    and W are THIS rank's contiguous block of rows; H is replicated.
    """

    _EPS = 10 ** -8          # nmf.py:69
    _ALGO = _lib.ALGO_NMF
    _REBIND_W = False        # SNMF rebinds self.W (snmf.py:70); NMF mutates in place
    _SHIPPED = True          # marks the classes whose hooks are the built-in C calls
    _HOOKS = ("update_w", "update_h", "frobenius_norm", "converged")
    #: rows per tile for out-of-core data: when set (or env PYMF_STREAM_ROWS), `data` is never made
    #: resident -- every iteration reads it tile by tile through `data[r0:r1, :]` (an h5py dataset,
    #: a np.memmap, anything with that slicing) and streams the tiles through the device.
    stream_rows = None
    #: True: before every factorize() / single hook call the bytes of `data` are digested and V is
    #: uploaded again when they changed (the reference reads self.data[:,:] afresh every time).
    #: False: `data` is uploaded once per object; call invalidate_data() after editing it in place.
    check_data = True
    #: W and H live on the device between calls.  The reference updates the arrays the caller may be
    #: holding IN PLACE (`w = mdl.W; mdl.factorize(); w` has changed, nmf.py:125-126,131-132), so after
    #: every call the host arrays are brought up to date -- unless nobody but this object can see them:
    #: an array that the object owns outright (CPython reference count: no other name, no view of it
    #: alive) cannot be observed between two calls, and it is refreshed when `.W` / `.H` is next read
    #: instead (768 MiB less over PCIe and through the float64 conversion per call at 1 048 576 x 256,
    #: k = 64).  The same test spares the digest of an array that was not handed out since it was last
    #: made equal to the device copy.  True: refresh after every call, digest before every call.
    eager_factors = False

    def __init__(self, data, num_bases=4):
        def setup_logging():                                   # nmf.py:73-90
            self._logger = logging.getLogger("pymf")
            if len(self._logger.handlers) < 1:
                ch = logging.StreamHandler()
                ch.setLevel(logging.DEBUG)
                ch.setFormatter(logging.Formatter("%(asctime)s [%(levelname)s] %(message)s"))
                self._logger.addHandler(ch)

        setup_logging()
        self.data = data                                       # nmf.py:93 (by reference)
        self._num_bases = num_bases                            # nmf.py:94
        (self._data_dimension, self._num_samples) = self.data.shape   # nmf.py:97
        self._ctx = None
        self._v_src = None       # the `data` object currently resident on the device
        self._v_fp = None        # ... and the digest of its bytes at upload time
        self._w_fp = None        # digest of the host W/H the device copies equal
        self._h_fp = None
        self._host_stale = set() # factors whose DEVICE copy is newer than the host array
        self._handed = set()     # factors whose host array was read through .W / .H since it was last synchronised
        self._defer_pull = False
        self._in_loop = False
        self._loop_data_checked = False
        #: wall-clock ms of the last factorize() call's parts: ctx (context creation), init (lazy init_w / init_h),
        #: upload (digests + host -> device copies of data, W, H), loop (device loop), total
        self.last_call_ms = {}

    def _tick(self, key, t0):
        d = self.__dict__.setdefault("last_call_ms", {})
        d[key] = d.get(key, 0.0) + (time.perf_counter() - t0) * 1e3

    # ---- W / H: plain attributes to the user, lazily refreshed from the device in the hook loop ----
    def _factor_get(self, name):
        try:
            arr = self.__dict__["_" + name]
        except KeyError:
            raise AttributeError("'%s' object has no attribute '%s'" % (type(self).__name__, name))
        if name in self.__dict__.get("_host_stale", ()):
            self._refresh_host(name)
            arr = self.__dict__["_" + name]
        self.__dict__.setdefault("_handed", set()).add(name)   # the caller may edit it in place from now on
        return arr

    def _factor_set(self, name, value):
        self.__dict__["_" + name] = value
        self.__dict__.setdefault("_host_stale", set()).discard(name)
        self.__dict__["_%s_fp" % name.lower()] = None          # host is newer: upload at the next call

    def _has(self, name):
        """hasattr(self, 'W') without reading the property (a read refreshes the host array)."""
        return ("_" + name) in self.__dict__

    def _held_elsewhere(self, name):
        """May anybody but this object see (or edit) the host array of W / H?  False only for a plain
        ndarray that owns its memory and has no reference besides this object's (CPython refcount: the
        attribute itself and the argument of getrefcount -- a name the caller kept, or a live view,
        shows up as a third)."""
        d = self.__dict__
        key = "_" + name
        if self.eager_factors or not hasattr(sys, "getrefcount"):
            return True
        if type(d[key]) is not np.ndarray or d[key].base is not None or not d[key].flags.owndata:
            return True
        return sys.getrefcount(d[key]) > 2

    def _factor_del(self, name):
        try:
            del self.__dict__["_" + name]
        except KeyError:
            raise AttributeError(name)
        self.__dict__.setdefault("_host_stale", set()).discard(name)

    W = property(lambda self: self._factor_get("W"), lambda self, v: self._factor_set("W", v),
                 lambda self: self._factor_del("W"))
    H = property(lambda self: self._factor_get("H"), lambda self, v: self._factor_set("H", v),
                 lambda self: self._factor_del("H"))

    # ---- device plumbing ---------------------------------------------------------
    def _world(self):
        return _dist.world()

    def _row_span(self):
        """(first global row, one past the last, rows of the whole matrix) of this rank's block:
        `data` holds the local rows, the global picture comes from one exchange of the row counts."""
        w = self._world()
        if w.size == 1:
            return 0, self._data_dimension, self._data_dimension
        if getattr(self, "_span", None) is None:
            counts = _dist.allgather_int(self._data_dimension)
            lo = int(sum(counts[:w.rank]))
            self._span = (lo, lo + int(counts[w.rank]), int(sum(counts)))
        return self._span

    def _global_rows(self):
        """Rows of the whole matrix: under a multi-rank world `data` is this rank's block."""
        return self._row_span()[2]

    def _context(self):
        if self._ctx is None:
            t0 = time.perf_counter()
            # one rank: a plain context; more: RCCL communicator (or, for ranks that cannot form one -- e.g. sharing a GPU --
            # the host / one-shot IPC transports) set up and self-tested under a time-box: pymf_amd.dist.make_context
            self._ctx = _dist.make_context(self._ALGO, self._data_dimension, self._num_samples, self._num_bases)
            self._tick("ctx", t0)
        return self._ctx

    def invalidate_data(self):
        """Tell the object that `data` was edited in place (needed only with check_data = False)."""
        self._v_src = None
        self._v_fp = None
        if self._ctx is not None:
            self._ctx.invalidate_v()

    def _upload_data(self, ctx):
        same_obj = self._v_src is self.data
        if same_obj and not self.check_data:
            return
        if _is_sparse(self.data):
            csr = self.data.tocsr()
            fp = tuple(_fingerprint(x) for x in (csr.indptr, csr.indices, csr.data)) if self.check_data else None
            if same_obj and fp == self._v_fp:
                return
            self._uploaded = True
            self._upload_sparse(ctx)
        else:
            arr = np.asarray(self.data[:, :])                   # data[:,:] idiom, nmf.py:123,129
            self._warn_if_float64(arr)
            fp = _fingerprint(arr) if self.check_data else None
            if same_obj and fp == self._v_fp:
                return
            self._uploaded = True
            ctx.set_v_dense(arr)
        self._v_src = self.data
        self._v_fp = fp

    def _warn_if_float64(self, arr):
        if getattr(arr, "dtype", None) == np.float64 and not self.__dict__.get("_warned_f64", False):
            self._warned_f64 = True
            warnings.warn("%s: float64 data is rounded to float32 on the device (fp32 MFMA arithmetic; the "
                          "reference would compute in float64, nmf.py:122-132) -- tolerances: DESIGN.md section 4"
                          % type(self).__name__, PrecisionWarning, stacklevel=6)

    def _upload_sparse(self, ctx):
        raise TypeError("scipy.sparse data is not supported by %s (the reference fails "
                        "with UFuncTypeError at nmf.py:131)" % type(self).__name__)

    def _stream_rows(self):
        """Tile height of the streamed mode (rounded to 64 rows), or 0 when `data` is kept resident."""
        r = self.stream_rows or int(os.environ.get("PYMF_STREAM_ROWS", "0") or 0)
        if not r or self._ALGO == _lib.ALGO_RNMF or _is_sparse(self.data):
            return 0
        return max(64, (int(r) + 63) // 64 * 64)

    def _stream_pass(self, ctx, rows, **flags):
        """One pass over `data` in row tiles (pmf_stream_begin / _tile / _end)."""
        ctx.stream_begin(max_tile_rows=rows, **flags)
        m = self._data_dimension
        for r0 in range(0, m, rows):
            tile = np.asarray(self.data[r0:min(m, r0 + rows), :])
            self._warn_if_float64(tile)
            ctx.stream_tile(r0, tile)
        return ctx.stream_end()

    def _stream_iteration(self, ctx, rows, compute_w, compute_h, compute_err):
        ferr, needs_direct = self._stream_pass(ctx, rows, compute_w=compute_w, compute_h=compute_h,
                                               compute_err=compute_err)
        if compute_err and needs_direct:          # nearly exact fit: the trace identity cancels
            ferr, _ = self._stream_pass(ctx, rows, resid=True)
        return ferr

    def _sync_to_device(self, with_data=True):
        ctx = self._context()
        t0 = time.perf_counter()
        try:
            self._uploaded = False
            try:
                ctx = self._sync_to_device_timed(ctx, with_data)
            except BaseException:
                self._uploaded = True             # (whatever was half done: the peers drop their derived state too)
                raise
            finally:
                if self._world().size > 1:
                    # Which cached sums are current ((W^T V | W^T W), ||V||^2, the trace terms) decides which COLLECTIVES the
                    # next call runs -- and a rank whose rows were not touched keeps its caches when another rank's data or W
                    # rows were edited: the ranks would then disagree about the next all-reduce (one waits for a peer that
                    # never comes).  So the ranks agree here: if anything was uploaded anywhere, everybody drops the derived
                    # state.  The vote runs in a `finally`: a rank whose upload RAISED (a bad dtype) still takes part, so its
                    # peers are not stranded.  Only the entry points that run collectives anyway come through here
                    # (factorize, update_w / update_h, frobenius_norm, update_s); reads of local state (RNMF.S, copies,
                    # pickles) do not -- a rank-0-only checkpoint must not become a collective.
                    if _dist.allreduce_sum_array(np.array([1.0 if self._uploaded else 0.0]), tag="uploaded")[0] > 0:
                        ctx.invalidate_v()
            return ctx
        finally:
            self._tick("upload", t0)

    def _sync_to_device_timed(self, ctx, with_data):
        if with_data and (not self._in_loop or not self._loop_data_checked):
            # inside factorize()'s hook loop `data` is checked once, by the first shipped hook that runs
            self._loop_data_checked = True
            if self._stream_rows():
                if self._v_src is not self.data or self.check_data:
                    self._uploaded = True
                    ctx.invalidate_v()            # streamed data is re-read every pass; ||V||^2 with it
                    self._v_src = self.data
            else:
                self._upload_data(ctx)
        for name, fp_attr, setter in (("W", "_w_fp", ctx.set_w), ("H", "_h_fp", ctx.set_h)):
            if name in self._host_stale:          # the device copy is the newer one
                continue
            if not self._has(name):
                getattr(self, name)               # AttributeError, as the reference's self.W would raise
            if (getattr(self, fp_attr) is not None and name not in self._handed
                    and not self._held_elsewhere(name)):
                continue                          # equal to the device copy and out of everybody's reach since
            arr = self.__dict__["_" + name]
            if not np.issubdtype(np.asarray(arr).dtype, np.floating):
                # reference: `W *= dot(...)` on an integer W raises UFuncTypeError
                raise TypeError("%s must be a floating-point array" % name)
            fp = _fingerprint(arr)
            if getattr(self, fp_attr) != fp:
                self._uploaded = True
                setter(arr)
                setattr(self, fp_attr, fp)
            del arr
            self._handed.discard(name)
        return ctx

    def _refresh_host(self, name):
        """Bring the host array of W or H up to date with the device (the deferred half of _pull)."""
        ctx = self._context()
        self._host_stale.discard(name)
        if name == "W":
            cur = self.__dict__["_W"]
            if self._REBIND_W:
                w = ctx.get_w()
                self.__dict__["_W"] = w.astype(cur.dtype, copy=False) if cur.dtype != np.float32 else w
            elif not (hasattr(ctx, "get_w_into") and ctx.get_w_into(cur)):   # in place, nmf.py:131-132 (float64: widened on the device)
                np.copyto(cur, ctx.get_w(), casting="same_kind")
            self._w_fp = _fingerprint(self.__dict__["_W"])
        else:
            cur = self.__dict__["_H"]
            if not (hasattr(ctx, "get_h_into") and ctx.get_h_into(cur)):     # in place, nmf.py:125-126
                np.copyto(cur, ctx.get_h(), casting="same_kind")
            self._h_fp = _fingerprint(self.__dict__["_H"])

    def _pull(self, ctx, want_w, want_h):
        for name, want in (("W", want_w), ("H", want_h)):
            if not want:
                continue
            self._host_stale.add(name)
            if not self._defer_pull and self._held_elsewhere(name):
                self._refresh_host(name)          # somebody holds the array: the reference's in-place update

    def _flush_host(self, force=False):
        for name in sorted(self._host_stale):
            if force or self._held_elsewhere(name):
                self._refresh_host(name)

    def __getstate__(self):                       # pickling / copying: host arrays up to date, no device handle
        self._flush_host(force=True)
        st = dict(self.__dict__)
        st["_ctx"] = None
        st["_v_src"] = st["_v_fp"] = st["_w_fp"] = st["_h_fp"] = None
        st["_host_stale"], st["_handed"] = set(), set()       # the copy shares no bookkeeping with the original
        st["_defer_pull"] = st["_in_loop"] = st["_loop_data_checked"] = False
        st.pop("_span", None)
        return st

    # ---- reference surface ----------------------------------------------------------
    def frobenius_norm(self):
        """||data - W H||_F (nmf.py:100-114); -123456 without W/H or for sparse data."""
        if self._has('H') and self._has('W') and not _is_sparse(self.data):
            rows = self._stream_rows()
            if rows:
                return self._stream_pass(self._sync_to_device(), rows, resid=True)[0]
            return self._sync_to_device().frobenius()
        return -123456

    def init_w(self):                                          # nmf.py:116-117
        w = self._world()
        if w.size > 1:
            # every rank continues rank 0's stream: each draws the GLOBAL W0 and keeps its own rows,
            # and the H0 drawn afterwards is the same everywhere
            _dist.share_rng_state()
            lo, hi, mg = self._row_span()
            self.W = _draw_rows(mg, self._num_bases, lo, hi)
        else:
            self.W = np.random.random((self._data_dimension, self._num_bases))

    def init_h(self):                                          # nmf.py:119-120
        if self._world().size > 1:
            _dist.share_rng_state()                            # H is replicated: one stream, one H0
        self.H = np.random.random((self._num_bases, self._num_samples))

    def update_h(self):                                        # nmf.py:122-126
        ctx = self._sync_to_device()
        rows = self._stream_rows()
        if rows:
            self._stream_iteration(ctx, rows, False, True, False)
        else:
            ctx.update_h()
        self._pull(ctx, False, True)

    def update_w(self):                                        # nmf.py:128-132
        ctx = self._sync_to_device()
        snap = self._snapshot_w_if_the_step_may_fail(ctx)
        rows = self._stream_rows()
        try:
            if rows:
                self._stream_iteration(ctx, rows, True, False, False)
            else:
                ctx.update_w()
        except Exception:                                      # e.g. SNMF: LinAlgError behind a singular H H^T
            self._after_failed_w_step(ctx, snap)
            raise
        self._pull(ctx, True, False)

    #: True for classes whose W step can raise (SNMF: np.linalg.inv on a singular H H^T, snmf.py:69)
    _W_STEP_MAY_FAIL = False

    def _snapshot_w_if_the_step_may_fail(self, ctx, whole_loop=False):
        """The reference raises before it rebinds W (snmf.py:69-70): W is what it was, H is untouched.  A failing
        step here leaves garbage in the device W (and, behind a whole pmf_factorize loop, in the device H), so
        where that can happen and the device copy is the only current one (the host array was not refreshed
        since), W is copied device to device first and -- before a whole loop -- a device-only H (k x n: cheap)
        is brought to the host (a single W step does not touch H)."""
        if not self._W_STEP_MAY_FAIL:
            return False
        if whole_loop and "H" in self._host_stale:
            self._refresh_host("H")
        if "W" in self._host_stale:
            if hasattr(ctx, "snapshot_w"):
                ctx.snapshot_w()
                return True
            self._refresh_host("W")                            # a context without snapshots: the host keeps the copy
        return False

    def _after_failed_w_step(self, ctx, snap):
        if snap:
            ctx.restore_w()                                    # device W = the previous W again; still the newer copy
        elif "W" not in self._host_stale:
            self._w_fp = None                                  # the (current) host array goes up again with the next call
        # else: the device copy is the only current one and nothing better exists -- it stays the newer copy; a
        # stale host array is never declared current

    def _after_failed_call(self, ctx, snap):
        """A whole factorize() raised: the device factors are in an unknown state wherever the host arrays were
        current when the call started (they go up again); a factor that lived on the device alone keeps doing so."""
        self._after_failed_w_step(ctx, snap)
        if "H" not in self._host_stale:
            self._h_fp = None

    def converged(self, i):                                    # nmf.py:134-139
        derr = np.abs(self.ferr[i] - self.ferr[i - 1]) / self._num_samples
        return bool(derr < self._EPS)

    def _hooks_overridden(self):
        """True when update_w / update_h / frobenius_norm / converged are not the shipped class's own
        (a user subclass or an instance attribute): factorize() must then call them, as the
        reference's template-method loop does (nmf.py:182-202)."""
        cls = type(self)
        base = next((c for c in cls.__mro__ if c.__dict__.get("_SHIPPED", False)), None)
        if base is None:
            return True
        for name in self._HOOKS:
            if name in self.__dict__ or getattr(cls, name) is not getattr(base, name):
                return True
        return False

    def factorize(self, niter=1, show_progress=False,
                  compute_w=True, compute_h=True, compute_err=True):
        """Factorize s.t. WH = data (nmf.py:141-202)."""
        if show_progress:                                      # nmf.py:166-169
            self._logger.setLevel(logging.INFO)
        else:
            self._logger.setLevel(logging.ERROR)

        t_call = time.perf_counter()
        self.last_call_ms = {}
        if not self._has('W'):                                 # nmf.py:173-174
            self.init_w()
        if not self._has('H'):                                 # nmf.py:176-177
            self.init_h()
        self._tick("init", t_call)

        if compute_err:                                        # nmf.py:179-180
            self.ferr = np.zeros(niter)

        rows = self._stream_rows()
        if self._hooks_overridden() or (show_progress and not rows):
            # the hooks synchronise what THEY need: a subclass that computes on the host never pays for an upload
            return self._factorize_by_hooks(niter, compute_w, compute_h, compute_err)
        late = None if rows else self._late_data_check_ok(niter)
        ctx = self._sync_to_device(with_data=late is None)
        snap = compute_w and self._snapshot_w_if_the_step_may_fail(ctx, whole_loop=True)
        if late is not None:
            try:
                late = _LateDataCheck(self, ctx, late)         # W / H kept aside on the device, the digest of `data` under way
            except _lib.PmfError:                              # (no room for the copies: check first, as before)
                late = None
                self._upload_data(ctx)
        if rows:                                               # a Python loop already: logs as it runs
            ferr, done, conv_at = self._factorize_streamed(ctx, rows, niter, compute_w, compute_h, compute_err)
            self._last_iters = done
            self._pull(ctx, compute_w and done > 0, compute_h and done > 0)
            if compute_err:
                self.ferr[:done] = ferr[:done]
                if conv_at >= 0:                               # nmf.py:198-202
                    self.ferr = self.ferr[:conv_at]
            return
        else:
            try:
                try:
                    ferr, done, conv_at = ctx.factorize(niter, compute_w, compute_h, compute_err,
                                                        conv_eps=self._EPS)
                except Exception:
                    if late is None or not late.data_changed():
                        raise
                if late is not None and late.data_changed():
                    # `data` was edited in place since its upload: the loop above ran on the OLD bytes (it was asked to stop
                    # as soon as the digest knew).  W and H go back to what they were, the new bytes go up, and the call
                    # starts again -- what the reference computes, which reads self.data[:,:] afresh (nmf.py:123,129)
                    late.restart()
                    ferr, done, conv_at = ctx.factorize(niter, compute_w, compute_h, compute_err,
                                                        conv_eps=self._EPS)
            except Exception:
                # the device factors are in an unknown state (e.g. behind a singular H H^T): W goes back to the
                # snapshot (or to the host array where that is the current one), H to the host array (a class
                # whose step may fail had it flushed before the call)
                self._after_failed_call(ctx, snap)
                raise
            finally:
                # however this call ends (KeyboardInterrupt between the loop and the digest's verdict, an error re-raised from
                # the side thread): the digest thread has finished with `data` and the context's abort request is withdrawn --
                # a flag left set would make the NEXT pmf_factorize return at once with iters_done = 0 and no error (round-5 advisor)
                if late is not None:
                    late.finish()
        self._last_iters = done
        self._pull(ctx, compute_w and done > 0, compute_h and done > 0)
        if hasattr(ctx, "last_loop_ms"):
            self.last_call_ms["loop"] = ctx.last_loop_ms()
        self._tick("total", t_call)

        for i in range(done):                                  # nmf.py:189-194
            if compute_err:
                self.ferr[i] = ferr[i]
                self._logger.info('Iteration ' + str(i + 1) + '/' + str(niter) +
                                  ' FN:' + str(self.ferr[i]))
            else:
                self._logger.info('Iteration ' + str(i + 1) + '/' + str(niter))
        if compute_err and conv_at >= 0:                       # nmf.py:198-202
            self.ferr = self.ferr[:conv_at]

    #: The digest of a large `data` before every factorize() (check_data) runs BESIDE the device loop instead of in front of
    #: it: the loop starts on the resident copy at once; if the bytes turn out to have changed since the upload the loop is
    #: stopped, W / H are put back from copies kept on the device, the new bytes go up and the call starts again
    #: (_LateDataCheck).  Same results as checking first -- the common case (nothing changed) no longer pays 4-6 ms per GiB.
    _LATE_DATA_CHECK = True
    _LATE_DATA_CHECK_MIN_BYTES = int(os.environ.get("PYMF_LATE_CHECK_MIN_BYTES", str(64 << 20)))   # (0: always -- the test sweeps)

    def _late_data_check_ok(self, niter):
        """The dense ndarray to digest beside the loop, or None where the data are checked in front of it as before: small
        data, first upload, another object bound to `data`, check_data off, several ranks (the ranks vote on uploads before
        the loop), classes with device state derived from data (RNMF) or a schedule that a restart would have to rewind (BNMF)."""
        if (not self._LATE_DATA_CHECK or not self.check_data or niter < 1 or self._world().size > 1
                or self._ALGO not in (_lib.ALGO_NMF, _lib.ALGO_SNMF, _lib.ALGO_NMFALS)
                or self._v_src is not self.data or self._v_fp is None or _is_sparse(self.data)
                or type(self.data) is not np.ndarray or self.data.nbytes < self._LATE_DATA_CHECK_MIN_BYTES
                or 2 * self._num_bases > self._num_samples       # (the device-side copy of W must be small beside `data`)
                or self._ctx is None or not all(hasattr(self._ctx, a) for a in ("abort", "snapshot_w", "snapshot_h", "restore_w", "restore_h"))):
            return None
        return self.data

    def _factorize_by_hooks(self, niter, compute_w, compute_h, compute_err):
        """The reference's loop (nmf.py:182-202), one hook call at a time, log lines as it runs.
        W and H stay on the device between the hooks; the host arrays are refreshed when a hook (or
        the user) reads `.W` / `.H`, and in any case before factorize() returns."""
        self._defer_pull = True
        self._in_loop = True
        self._loop_data_checked = False
        done = 0
        try:
            for i in range(niter):
                if compute_w:
                    self.update_w()
                if compute_h:
                    self.update_h()
                done = i + 1
                if compute_err:
                    self.ferr[i] = self.frobenius_norm()
                    self._logger.info('Iteration ' + str(i + 1) + '/' + str(niter) +
                                      ' FN:' + str(self.ferr[i]))
                else:
                    self._logger.info('Iteration ' + str(i + 1) + '/' + str(niter))
                if i > 1 and compute_err:
                    if self.converged(i):
                        self.ferr = self.ferr[:i]
                        break
        finally:
            self._defer_pull = False
            self._in_loop = False
            self._last_iters = done
            self._flush_host()

    def _factorize_streamed(self, ctx, rows, niter, compute_w, compute_h, compute_err):
        """The loop of nmf.py:182-202 with one streamed pass per iteration; same return triple as
        Context.factorize: (ferr, iterations executed, iteration at which converged() fired or -1)."""
        ferr = np.zeros(max(int(niter), 1)) if compute_err else None
        done, conv_at = 0, -1
        for i in range(niter):
            e = self._stream_iteration(ctx, rows, compute_w, compute_h, compute_err)
            done = i + 1
            if compute_err:
                ferr[i] = e
                self._logger.info('Iteration ' + str(i + 1) + '/' + str(niter) + ' FN:' + str(ferr[i]))
                if i > 1 and abs(ferr[i] - ferr[i - 1]) / self._num_samples < self._EPS:   # nmf.py:134-139,198
                    conv_at = i
                    break
            else:
                self._logger.info('Iteration ' + str(i + 1) + '/' + str(niter))
        return ferr, done, conv_at
