"""pymf_amd.NMF -- drop-in for pymf.NMF (reference pymf/nmf.py) on MI355X.

Same class surface as the reference (`NMF(data, num_bases=4)`,
`.factorize(niter, show_progress, compute_w, compute_h, compute_err)`, `.W`,
`.H`, `.ferr`, `.frobenius_norm()`, the overridable hooks `init_w / init_h /
update_w / update_h / converged`, `_EPS`), but every update runs as HIP kernels
in libpymf_hip.so through a ctypes C ABI: `factorize()` is ONE C call.
There is no NumPy fallback: without the library (or a GPU) the calls raise.

Host-visible state follows the reference: W/H are created lazily, W first
(nmf.py:173-177), default float64 (nmf.py:117,120); NMF writes results back IN
PLACE into the existing arrays (nmf.py:125-126,131-132) keeping their dtype;
the arithmetic on the device is float32.
"""
import logging
import os

import numpy as np

from . import _lib
from . import dist as _dist

__all__ = ["NMF"]


def _is_sparse(x):
    try:
        import scipy.sparse
        return scipy.sparse.issparse(x)
    except Exception:        # pragma: no cover
        return False


def _fingerprint(a):
    """Change detector for a host array: identity, layout and the float64 sum of ALL elements (one
    pass at memory speed, cheaper than the float32 conversion + upload it saves).  A single-element
    poke between two factorize() calls changes the sum, so the device copy is refreshed -- the
    reference always computes from the current host arrays."""
    flat = np.asarray(a).reshape(-1)
    return (id(a), a.shape, a.dtype.str, float(np.sum(flat, dtype=np.float64)),
            float(flat[-1]) if flat.shape[0] else 0.0)


class NMF(object):
    """Non-negative matrix factorization, multiplicative updates (Lee & Seung).

    Parameters mirror the reference (pymf/nmf.py:23-66): data is m x n
    (m = _data_dimension rows, n = _num_samples columns), W is m x num_bases,
    H is num_bases x n.
    """

    _EPS = 10 ** -8          # nmf.py:69
    _ALGO = _lib.ALGO_NMF
    _REBIND_W = False        # SNMF rebinds self.W (snmf.py:70); NMF mutates in place
    #: rows per tile for out-of-core data: when set (or env PYMF_STREAM_ROWS), `data` is never made
    #: resident -- every iteration reads it tile by tile through `data[r0:r1, :]` (an h5py dataset,
    #: a np.memmap, anything with that slicing) and streams the tiles through the device.
    stream_rows = None

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
        self._w_fp = None        # fingerprint of the host W/H the device copies equal
        self._h_fp = None

    # ---- device plumbing ---------------------------------------------------------
    def _world(self):
        return _dist.world()

    def _global_rows(self):
        """Rows of the whole matrix: under a multi-rank world `data` is this rank's block."""
        return getattr(self, "_m_global", self._data_dimension)

    def _context(self):
        if self._ctx is None:
            w = self._world()
            self._ctx = _lib.Context(self._ALGO, self._data_dimension, self._num_samples,
                                     self._num_bases, device=w.local_rank, rank=w.rank,
                                     nranks=w.size, nccl_id=w.nccl_id)
        return self._ctx

    def _upload_data(self, ctx):
        if self._v_src is self.data:
            return
        if _is_sparse(self.data):
            self._upload_sparse(ctx)
        else:
            ctx.set_v_dense(np.asarray(self.data[:, :]))        # data[:,:] idiom, nmf.py:123,129
        self._v_src = self.data

    def _upload_sparse(self, ctx):
        raise TypeError("scipy.sparse data is not supported by %s (the reference fails "
                        "with UFuncTypeError at nmf.py:131)" % type(self).__name__)

    def _stream_rows(self):
        """Tile height of the streamed mode (rounded to 64 rows), or 0 when `data` is kept resident."""
        r = self.stream_rows or int(os.environ.get("PYMF_STREAM_ROWS", "0") or 0)
        if not r or self._ALGO != _lib.ALGO_NMF or _is_sparse(self.data):
            return 0
        return max(64, (int(r) + 63) // 64 * 64)

    def _stream_pass(self, ctx, rows, **flags):
        """One pass over `data` in row tiles (pmf_stream_begin / _tile / _end)."""
        ctx.stream_begin(max_tile_rows=rows, **flags)
        m = self._data_dimension
        for r0 in range(0, m, rows):
            ctx.stream_tile(r0, np.asarray(self.data[r0:min(m, r0 + rows), :]))
        return ctx.stream_end()

    def _stream_iteration(self, ctx, rows, compute_w, compute_h, compute_err):
        ferr, needs_direct = self._stream_pass(ctx, rows, compute_w=compute_w, compute_h=compute_h,
                                               compute_err=compute_err)
        if compute_err and needs_direct:          # nearly exact fit: the trace identity cancels
            ferr, _ = self._stream_pass(ctx, rows, resid=True)
        return ferr

    def _sync_to_device(self, with_data=True):
        ctx = self._context()
        if with_data and not self._stream_rows():
            self._upload_data(ctx)
        for name, fp_attr, setter in (("W", "_w_fp", ctx.set_w), ("H", "_h_fp", ctx.set_h)):
            arr = getattr(self, name)
            if not np.issubdtype(np.asarray(arr).dtype, np.floating):
                # reference: `W *= dot(...)` on an integer W raises UFuncTypeError
                raise TypeError("%s must be a floating-point array" % name)
            fp = _fingerprint(arr)
            if getattr(self, fp_attr) != fp:
                setter(arr)
                setattr(self, fp_attr, fp)
        return ctx

    def _pull(self, ctx, want_w, want_h):
        if want_w:
            w = ctx.get_w()
            if self._REBIND_W:
                self.W = w.astype(self.W.dtype, copy=False) if self.W.dtype != np.float32 else w
            else:
                np.copyto(self.W, w, casting="same_kind")       # in place, nmf.py:131-132
            self._w_fp = _fingerprint(self.W)
        if want_h:
            np.copyto(self.H, ctx.get_h(), casting="same_kind")  # in place, nmf.py:125-126
            self._h_fp = _fingerprint(self.H)

    # ---- reference surface ----------------------------------------------------------
    def frobenius_norm(self):
        """||data - W H||_F (nmf.py:100-114); -123456 without W/H or for sparse data."""
        if hasattr(self, 'H') and hasattr(self, 'W') and not _is_sparse(self.data):
            rows = self._stream_rows()
            if rows:
                return self._stream_pass(self._sync_to_device(), rows, resid=True)[0]
            return self._sync_to_device().frobenius()
        return -123456

    def init_w(self):                                          # nmf.py:116-117
        w = self._world()
        if w.size > 1:
            lo, hi = w.row_range(self._global_rows())
            self.W = np.random.random((self._global_rows(), self._num_bases))[lo:hi].copy()
        else:
            self.W = np.random.random((self._data_dimension, self._num_bases))

    def init_h(self):                                          # nmf.py:119-120
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
        rows = self._stream_rows()
        if rows:
            self._stream_iteration(ctx, rows, True, False, False)
        else:
            ctx.update_w()
        self._pull(ctx, True, False)

    def converged(self, i):                                    # nmf.py:134-139
        derr = np.abs(self.ferr[i] - self.ferr[i - 1]) / self._num_samples
        return bool(derr < self._EPS)

    def factorize(self, niter=1, show_progress=False,
                  compute_w=True, compute_h=True, compute_err=True):
        """Factorize s.t. WH = data (nmf.py:141-202); one call into libpymf_hip."""
        if show_progress:                                      # nmf.py:166-169
            self._logger.setLevel(logging.INFO)
        else:
            self._logger.setLevel(logging.ERROR)

        if not hasattr(self, 'W'):                             # nmf.py:173-174
            self.init_w()
        if not hasattr(self, 'H'):                             # nmf.py:176-177
            self.init_h()

        if compute_err:                                        # nmf.py:179-180
            self.ferr = np.zeros(niter)

        ctx = self._sync_to_device()
        rows = self._stream_rows()
        if rows:
            ferr, done, conv_at = self._factorize_streamed(ctx, rows, niter, compute_w, compute_h, compute_err)
        else:
            ferr, done, conv_at = ctx.factorize(niter, compute_w, compute_h, compute_err,
                                                conv_eps=self._EPS)
        self._last_iters = done
        self._pull(ctx, compute_w and done > 0, compute_h and done > 0)

        for i in range(done):                                  # nmf.py:189-194
            if compute_err:
                self.ferr[i] = ferr[i]
                self._logger.info('Iteration ' + str(i + 1) + '/' + str(niter) +
                                  ' FN:' + str(self.ferr[i]))
            else:
                self._logger.info('Iteration ' + str(i + 1) + '/' + str(niter))
        if compute_err and conv_at >= 0:                       # nmf.py:198-202
            self.ferr = self.ferr[:conv_at]


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
                if i > 1 and abs(ferr[i] - ferr[i - 1]) / self._num_samples < self._EPS:   # nmf.py:134-139,198
                    conv_at = i
                    break
        return ferr, done, conv_at


def _setup_module():      # keep `python -m doctest`-style entry harmless
    return None
