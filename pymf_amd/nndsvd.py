"""pymf_amd.NNDSVD -- drop-in for pymf.NNDSVD (reference pymf/nndsvd.py) on MI355X  (SURVEY 8(f) row 4).

Non-negative double SVD (Boutsidis & Gallopoulos 2008): the deterministic initialiser the reference
offers for NMF -- factorize() fills W and H once, and the user copies them into an NMF model
(nndsvd.py:56-64).  The whole computation is one C call, pmf_nndsvd_init (Gram matrix and
U = data V S^-1 on fp32 MFMA, a float64 Jacobi eigensolver on the device, and the closed form of
the reference's per-basis second SVD; pymf_amd/csrc/pmf_nndsvd.h; beyond 1024 columns the k largest
eigenpairs by a Chebyshev-filtered subspace iteration on the float64 MFMA, pymf_amd/csrc/pmf_topk.h).

The reference's SVD works on data^T data when rows > cols and on data data^T otherwise
(svd.py:237-246).  The device routine takes the first form, so a wide matrix is passed transposed
(W and H swap roles and are transposed back) -- the same switch, made on the host.  Limit of this
build: min(rows, cols) <= 16384.
"""
import logging

import numpy as np

from . import _lib
from .nmf import NMF, _fingerprint

__all__ = ["NNDSVD"]


class NNDSVD(NMF):
    _SHIPPED = True
    def init_w(self):                                           # nndsvd.py:69-70
        self.W = np.zeros((self._data_dimension, self._num_bases))

    def init_h(self):                                           # nndsvd.py:72-73
        self.H = np.zeros((self._num_bases, self._num_samples))

    def update_h(self):                                         # nndsvd.py:75-76
        pass

    def update_w(self):                                         # nndsvd.py:78-106 (sets W AND H)
        w = self._world()
        tall = self._data_dimension > self._num_samples or w.size > 1
        if tall:
            ctx = self._context()
            self._upload_data(ctx)
            ctx.nndsvd_init()
            np.copyto(self.W, ctx.get_w(), casting="same_kind")
            np.copyto(self.H, ctx.get_h(), casting="same_kind")
            self._w_fp = _fingerprint(self.W)
            self._h_fp = _fingerprint(self.H)
        else:
            ctx = _lib.Context(_lib.ALGO_NMF, self._num_samples, self._data_dimension,
                               self._num_bases, device=w.local_rank)
            try:
                ctx.set_v_dense(np.ascontiguousarray(np.asarray(self.data[:, :]).T))
                ctx.nndsvd_init()
                np.copyto(self.W, ctx.get_h().T, casting="same_kind")
                np.copyto(self.H, ctx.get_w().T, casting="same_kind")
            finally:
                ctx.close()

    def factorize(self, niter=1, show_progress=False,
                  compute_w=True, compute_h=True, compute_err=True):
        """One pass, whatever niter / compute_w / compute_h say (nndsvd.py:108-114)."""
        if show_progress:                                       # nmf.py:166-169
            self._logger.setLevel(logging.INFO)
        else:
            self._logger.setLevel(logging.ERROR)
        if not self._has('W'):
            self.init_w()
        if not self._has('H'):
            self.init_h()
        if compute_err:
            self.ferr = np.zeros(1)
        self.update_w()
        self.update_h()
        if compute_err:
            self.ferr[0] = self.frobenius_norm()
            self._logger.info('Iteration 1/1 FN:' + str(self.ferr[0]))
        else:
            self._logger.info('Iteration 1/1')
