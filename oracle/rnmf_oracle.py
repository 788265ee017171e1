"""NumPy restatement of pymf.RNMF (reference pymf/rnmf.py:70-116) -- test oracle ("next" row 3).

Robust NMF: data ~ W H + S with a sparse outlier matrix S kept by soft thresholding
(rnmf.py:75-79,96-98).  The lazy-init quirks are part of the behaviour: init_h sets H = 1,
normalises the columns of W, scales the rows of H and creates S (rnmf.py:84-94); S exists only
after init_h / update_s, so pre-assigning BOTH W and H makes update_w fail exactly as the
reference does (AttributeError on self.S).
"""
import numpy as np
from .nmf_oracle import NMFOracle


def soft_thresholding(X, lamb):                       # rnmf.py:75-79
    X = np.where(np.abs(X) <= lamb, 0.0, X)
    X = np.where(X > lamb, X - lamb, X)
    X = np.where(X < -1.0 * lamb, X + lamb, X)
    return X


class RNMFOracle(NMFOracle):
    def __init__(self, data, num_bases=4, lamb=2.0):  # rnmf.py:70-73
        NMFOracle.__init__(self, data, num_bases=num_bases)
        self._lamb = lamb

    def init_h(self):                                 # rnmf.py:84-94
        self.H = np.random.random((self._num_bases, self._num_samples))
        self.H[:, :] = 1.0
        Wnorm = np.sqrt(np.sum(self.W ** 2.0, axis=0))
        self.W /= Wnorm
        for i in range(self.H.shape[0]):
            self.H[i, :] *= Wnorm[i]
        self.update_s()

    def update_s(self):                               # rnmf.py:96-98
        self.S = self.data - np.dot(self.W, self.H)
        self.S = soft_thresholding(self.S, self._lamb)

    def update_h(self):                               # rnmf.py:100-107
        H1 = np.dot(self.W.T, self.S - self.data)
        H1 = np.abs(H1) - H1
        H1 /= (2.0 * np.dot(self.W.T, np.dot(self.W, self.H)))
        self.H *= H1
        self.update_s()

    def update_w(self):                               # rnmf.py:109-115
        W1 = np.dot(self.S - self.data, self.H.T)
        W1 = np.abs(W1) - W1
        W1 /= (2.0 * (np.dot(self.W, np.dot(self.H, self.H.T))))
        self.W *= W1
