"""pymf_amd.BNMF -- drop-in for pymf.BNMF (reference pymf/bnmf.py) on MI355X  (SURVEY 8(f) row 1).

Binary matrix factorization: the NMF multiplicative updates with a penalty that pulls W and H
towards {0, 1} (bnmf.py:79-90):
    H *= (W^T V + 3 l_H H^2) / ((W^T W) H + 2 l_H H^3 + l_H H + 1e-9)
    W *= (V H^T + 3 l_W W^2) / (W (H H^T) + 2 l_W W^3 + l_W W + 1e-9)
l_W = l_H = 1/niter at the start of every factorize() (bnmf.py:118-119) and BOTH grow by 1.1 at the
end of every update_h (bnmf.py:84-85).  Same contractions as NMF, so it runs on the same fused
one-pass kernel with a different epilogue.
"""
from . import _lib
from .nmf import NMF

__all__ = ["BNMF"]


class BNMF(NMF):
    _SHIPPED = True
    _ALGO = _lib.ALGO_BNMF
    _LAMB_INCREASE_W = 1.1       # bnmf.py:76
    _LAMB_INCREASE_H = 1.1       # bnmf.py:77

    def _push_lambda(self):
        # self._lamb_W / _lamb_H do not exist before the first factorize(): AttributeError, as in
        # the reference (bnmf.py:80,88 read them unconditionally)
        self._context().set_lambda(self._lamb_W, self._lamb_H)

    def _pull_lambda(self):
        self._lamb_W, self._lamb_H = self._context().get_lambda()

    def update_h(self):
        self._push_lambda()
        NMF.update_h(self)
        self._pull_lambda()

    def update_w(self):
        self._push_lambda()
        NMF.update_w(self)

    def factorize(self, niter=10, compute_w=True, compute_h=True,
                  show_progress=False, compute_err=True):
        """bnmf.py:92-123 (note the reference's argument order differs from NMF.factorize)."""
        self._lamb_W = 1.0 / niter                 # bnmf.py:118-119
        self._lamb_H = 1.0 / niter
        self._push_lambda()
        NMF.factorize(self, niter=niter, compute_w=compute_w, compute_h=compute_h,
                      show_progress=show_progress, compute_err=compute_err)
        self._pull_lambda()
