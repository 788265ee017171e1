"""NumPy restatement of pymf.BNMF (reference pymf/bnmf.py:79-123) -- test oracle ("next" row 1)."""
import numpy as np
from .nmf_oracle import NMFOracle, EPS_DEN


class BNMFOracle(NMFOracle):
    _LAMB_INCREASE_W = 1.1      # bnmf.py:76
    _LAMB_INCREASE_H = 1.1      # bnmf.py:77

    def update_h(self):         # bnmf.py:79-85
        H, W, V = self.H, self.W, self.data[:, :]
        H1 = np.dot(W.T, V) + 3.0 * self._lamb_H * (H ** 2)
        H2 = np.dot(np.dot(W.T, W), H) + 2 * self._lamb_H * (H ** 3) + self._lamb_H * H + EPS_DEN
        H *= H1 / H2
        self._lamb_W = self._LAMB_INCREASE_W * self._lamb_W      # both schedules advance here
        self._lamb_H = self._LAMB_INCREASE_H * self._lamb_H

    def update_w(self):         # bnmf.py:87-90
        H, W, V = self.H, self.W, self.data[:, :]
        W1 = np.dot(V, H.T) + 3.0 * self._lamb_W * (W ** 2)
        W2 = np.dot(W, np.dot(H, H.T)) + 2.0 * self._lamb_W * (W ** 3) + self._lamb_W * W + EPS_DEN
        W *= W1 / W2

    def factorize(self, niter=10, compute_w=True, compute_h=True, compute_err=True):
        self._lamb_W = 1.0 / niter                                # bnmf.py:118-119
        self._lamb_H = 1.0 / niter
        NMFOracle.factorize(self, niter=niter, compute_w=compute_w, compute_h=compute_h,
                            compute_err=compute_err)
