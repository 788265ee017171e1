"""pymf_amd.RNMF -- drop-in for pymf.RNMF (reference pymf/rnmf.py) on MI355X  (SURVEY 8(f) row 3).

Robust NMF: data ~ W H + S, S a sparse outlier matrix kept by soft thresholding with threshold
`lamb` (rnmf.py:75-79,96-98).  Like the reference class it is not star-exported by the package.
Lazy initialisation is part of the behaviour and is reproduced on the host (rnmf.py:81-94):
init_h sets H = 1, normalises the columns of W, scales the rows of H and creates S; S exists only
after init_h / update_s, so assigning BOTH W and H before factorize() makes update_w fail with
AttributeError exactly as the reference does.
On the device the state is D = S - data, the matrix both contractions use (rnmf.py:102,111).
"""
import numpy as np

from . import _lib
from . import dist
from .nmf import NMF

__all__ = ["RNMF"]


class RNMF(NMF):
    _ALGO = _lib.ALGO_RNMF
    _SHIPPED = True

    def __init__(self, data, num_bases=4, lamb=2.0):            # rnmf.py:70-73
        NMF.__init__(self, data, num_bases=num_bases)
        self._lamb = lamb
        self._has_s = False

    def soft_thresholding(self, X, lamb):
        """The shrinkage operator of rnmf.py:75-79: entries within [-lamb, lamb] become 0, the others
        move towards 0 by lamb.  Host helper for users; the device thresholds inside `k_resid`."""
        X = np.asarray(X)
        return np.sign(X) * np.maximum(np.abs(X) - lamb, 0.0)

    @property
    def S(self):
        if not self._has_s:
            raise AttributeError("'RNMF' object has no attribute 'S'")   # rnmf.py: S is created by update_s
        # LOCAL: no upload, no vote.  The device keeps D = S - data against the data IT holds, so S = D + V(device) is the
        # attribute's value whatever happened to the host's `data` / W / H since (the reference's S is a plain attribute
        # that only update_s changes) -- and a rank-0-only `model.S` / pickle must not be a collective (advisor, round 4)
        pend = self.__dict__.get("_s_host")
        if pend is not None or self._ctx is None:
            if pend is None:
                raise AttributeError("'RNMF' object has no attribute 'S'")
            return pend                                          # a copy / unpickled object whose context has not seen S yet
        return self._ctx.rnmf_get_s()

    # The reference's S is an attribute: a copy or a pickle of the object carries it, and new `data` leaves it as it is
    # (update_w / update_h then work on S - new data until the next update_s).  Here S lives on the device as D = S - data:
    # a new V re-bases D inside pmf_set_v_* (no host traffic); a copy takes S along as a host array and hands it to its own
    # context the first time that context is used.
    def __getstate__(self):
        S = self.S if (self._has_s and self._ctx is not None) else self.__dict__.get("_s_host")
        st = NMF.__getstate__(self)
        st["_s_host"] = S
        return st

    def _sync_to_device_timed(self, ctx, with_data):
        ctx = NMF._sync_to_device_timed(self, ctx, True if self.__dict__.get("_s_host") is not None else with_data)
        S = self.__dict__.get("_s_host")
        if S is not None:                                       # a context that has not seen this object's S yet
            self._push_lambda()
            ctx.rnmf_set_s(S)
            self._s_host = None
        return ctx

    def init_h(self):                                           # rnmf.py:84-94
        # the reference draws a random H and overwrites it with ones (rnmf.py:85-86): the draw is kept
        # so that the global NumPy stream stays where the reference leaves it
        if self._world().size > 1:
            dist.share_rng_state()
        np.random.random((self._num_bases, self._num_samples))
        col_energy = np.einsum("ij,ij->j", self.W, self.W)
        if self._world().size > 1:                              # column norms over ALL ranks' rows
            col_energy = dist.allreduce_sum_array(col_energy)
        Wnorm = np.sqrt(col_energy)
        self.W /= Wnorm                                         # unit-norm bases, their scale moves into H
        self.H = np.ones((self._num_bases, self._num_samples)) * Wnorm[:, None]
        self.update_s()

    def _push_lambda(self):
        self._context().set_lambda(self._lamb, 0.0)

    def update_s(self):                                         # rnmf.py:96-98
        self._push_lambda()
        self._sync_to_device().rnmf_update_s()
        self._has_s = True

    def _require_s(self):
        if not self._has_s:
            raise AttributeError("'RNMF' object has no attribute 'S'")   # as rnmf.py:101,110

    def update_h(self):                                         # rnmf.py:100-107
        self._require_s()
        self._push_lambda()
        NMF.update_h(self)

    def update_w(self):                                         # rnmf.py:109-115
        self._require_s()
        self._push_lambda()
        NMF.update_w(self)

    def factorize(self, niter=1, show_progress=False,
                  compute_w=True, compute_h=True, compute_err=True):
        if not self._has('W'):                              # nmf.py:173-177 order: W, then H (+ S)
            self.init_w()
        if not self._has('H'):
            self.init_h()
        if niter > 0 and (compute_w or compute_h):
            self._require_s()
        self._push_lambda()
        NMF.factorize(self, niter=niter, show_progress=show_progress, compute_w=compute_w,
                      compute_h=compute_h, compute_err=compute_err)
