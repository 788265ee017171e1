"""pymf_amd.SNMF -- drop-in for pymf.SNMF (reference pymf/snmf.py) on MI355X.

Semi-NMF (Ding, Li, Jordan): data may be mixed-sign, W is unconstrained,
H >= 0.  update_w: W = (V H^T) inv(H H^T)  (snmf.py:67-70, REBINDS self.W);
update_h: the pos/neg-split multiplicative sqrt rule (snmf.py:72-91).
scipy.sparse CSR data is accepted (the reference cannot run on it; semantics =
dense SNMF on data.toarray(), frobenius_norm() keeps the reference's -123456).

Precision: `data` and W are float32 on the device like every other class, but H -- k x n, and the
operand of inv(H H^T), which amplifies its rounding by sigma_max / sigma_min of H (3e3 for a square
uniform H) -- is kept in FLOAT64 there, as the reference keeps it (nmf.py:120): a float64 self.H goes
up and comes back exactly, and the H step (snmf.py:72-91) runs on the float64 matrix cores
(num_bases <= 128).
"""
import numpy as np

from . import _lib
from .nmf import NMF, _is_sparse

__all__ = ["SNMF"]


class SNMF(NMF):
    _SHIPPED = True
    _ALGO = _lib.ALGO_SNMF
    _REBIND_W = True
    _W_STEP_MAY_FAIL = True      # np.linalg.inv(H H^T) raises on a singular matrix (snmf.py:69)

    def _upload_sparse(self, ctx):
        csr = self.data.tocsr()
        self._warn_if_float64(csr.data)
        csr.sum_duplicates()
        ctx.set_v_csr(csr.indptr, csr.indices, csr.data)

    def factorize(self, niter=1, show_progress=False,
                  compute_w=True, compute_h=True, compute_err=True):
        if _is_sparse(self.data) and compute_err:
            # reference: frobenius_norm() returns -123456 for sparse data (nmf.py:109-112),
            # so ferr is a constant and the loop "converges" at i == 2 (nmf.py:198-202).
            raise TypeError("compute_err=True on scipy.sparse data: the reference's "
                            "frobenius_norm() only returns its -123456 sentinel there; "
                            "pass compute_err=False")
        return NMF.factorize(self, niter=niter, show_progress=show_progress,
                             compute_w=compute_w, compute_h=compute_h, compute_err=compute_err)
