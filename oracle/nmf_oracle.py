"""NumPy restatement of pymf.NMF (reference pymf/nmf.py) -- test oracle.

Every function cites the reference lines it follows.  Operation order,
in-place semantics and dtype propagation are the reference's: the result
dtype follows the operands (float64 when W/H come from the default
np.random.random init, float32 only when V, W and H are all float32).
"""
import numpy as np

EPS_DEN = 10 ** -9      # added to denominators only (nmf.py:124,130)
EPS_CONV = 10 ** -8     # NMF._EPS (nmf.py:69)
SENTINEL = -123456      # frobenius_norm() without W/H (nmf.py:112)


def nmf_update_w(V, W, H):
    """nmf.py:128-132 -- W <- W * (V H^T) / ((W H) H^T + 1e-9), in place."""
    W2 = np.dot(np.dot(W, H), H.T) + EPS_DEN      # :130 (reference order (W H) H^T)
    W *= np.dot(V, H.T)                           # :131
    W /= W2                                       # :132
    return W


def nmf_update_h(V, W, H):
    """nmf.py:122-126 -- H <- H * (W^T V) / ((W^T W) H + 1e-9), in place."""
    H2 = np.dot(np.dot(W.T, W), H) + EPS_DEN      # :124
    H *= np.dot(W.T, V)                           # :125
    H /= H2                                       # :126
    return H


def _issparse(x):
    """scipy.sparse.issparse (nmf.py:109) without importing scipy for dense data: a sparse matrix is never an ndarray."""
    if isinstance(x, np.ndarray):
        return False
    try:
        import scipy.sparse
        return bool(scipy.sparse.issparse(x))
    except ImportError:
        return False


def frobenius_norm(V, W, H):
    """nmf.py:100-114 -- sqrt(sum((V - W H)^2))."""
    return np.sqrt(np.sum((V - np.dot(W, H)) ** 2))   # :110


class NMFOracle(object):
    """Driver restating NMF.__init__ / factorize (nmf.py:71-97, 141-202)."""

    update_w_fn = staticmethod(nmf_update_w)
    update_h_fn = staticmethod(nmf_update_h)
    rebinding_w = False

    def __init__(self, data, num_bases=4):
        self.data = data                                   # :93 by reference
        self._num_bases = num_bases                        # :94
        (self._data_dimension, self._num_samples) = data.shape   # :97

    def frobenius_norm(self):
        if hasattr(self, 'H') and hasattr(self, 'W') and not _issparse(self.data):   # :109 (both halves of the condition)
            return frobenius_norm(self.data[:, :], self.W, self.H)
        return SENTINEL                                    # :112

    def init_w(self):                                      # :116-117
        self.W = np.random.random((self._data_dimension, self._num_bases))

    def init_h(self):                                      # :119-120
        self.H = np.random.random((self._num_bases, self._num_samples))

    def update_w(self):
        r = type(self).update_w_fn(self.data[:, :], self.W, self.H)
        if type(self).rebinding_w:
            self.W = r

    def update_h(self):
        type(self).update_h_fn(self.data[:, :], self.W, self.H)

    def converged(self, i):                                # :134-139
        derr = np.abs(self.ferr[i] - self.ferr[i - 1]) / self._num_samples
        return bool(derr < EPS_CONV)

    def factorize(self, niter=1, compute_w=True, compute_h=True, compute_err=True):
        if not hasattr(self, 'W'):                         # :173-174 (W drawn first)
            self.init_w()
        if not hasattr(self, 'H'):                         # :176-177
            self.init_h()
        if compute_err:                                    # :179-180
            self.ferr = np.zeros(niter)
        for i in range(niter):                             # :182
            if compute_w:
                self.update_w()                            # :183-184
            if compute_h:
                self.update_h()                            # :186-187
            if compute_err:
                self.ferr[i] = self.frobenius_norm()       # :189-190
            if i > 1 and compute_err:                      # :198
                if self.converged(i):
                    self.ferr = self.ferr[:i]              # :201 (entry i dropped)
                    break
