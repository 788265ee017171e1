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
from .nmf import NMF

__all__ = ["RNMF"]


class RNMF(NMF):
    _ALGO = _lib.ALGO_RNMF

    def __init__(self, data, num_bases=4, lamb=2.0):            # rnmf.py:70-73
        NMF.__init__(self, data, num_bases=num_bases)
        self._lamb = lamb
        self._has_s = False

    def soft_thresholding(self, X, lamb):                       # rnmf.py:75-79 (host helper, as in the reference)
        X = np.where(np.abs(X) <= lamb, 0.0, X)
        X = np.where(X > lamb, X - lamb, X)
        X = np.where(X < -1.0 * lamb, X + lamb, X)
        return X

    @property
    def S(self):
        if not self._has_s:
            raise AttributeError("'RNMF' object has no attribute 'S'")   # rnmf.py: S is created by update_s
        return self._sync_to_device().rnmf_get_s()

    def init_h(self):                                           # rnmf.py:84-94
        self.H = np.random.random((self._num_bases, self._num_samples))
        self.H[:, :] = 1.0
        Wnorm = np.sqrt(np.sum(self.W ** 2.0, axis=0))
        if self._world().size > 1:                              # column norms over ALL ranks' rows
            from . import dist
            Wnorm = np.sqrt(dist.allreduce_sum_array(np.sum(self.W ** 2.0, axis=0)))
        self.W /= Wnorm
        for i in range(self.H.shape[0]):
            self.H[i, :] *= Wnorm[i]
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
        if not hasattr(self, 'W'):                              # nmf.py:173-177 order: W, then H (+ S)
            self.init_w()
        if not hasattr(self, 'H'):
            self.init_h()
        if niter > 0 and (compute_w or compute_h):
            self._require_s()
        self._push_lambda()
        NMF.factorize(self, niter=niter, show_progress=show_progress, compute_w=compute_w,
                      compute_h=compute_h, compute_err=compute_err)
