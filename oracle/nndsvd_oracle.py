"""NumPy restatement of pymf.NNDSVD (reference pymf/nndsvd.py:69-114) and of the dense part of
pymf.SVD it calls (pymf/svd.py:105-150,237-246) -- test oracle ("next" row 4).

TEST INFRASTRUCTURE ONLY.  Written in the reference's operation order, including the second SVD of
the positive part of every rank-one factor (nndsvd.py:92-106); `nndsvd_closed_form` is the
algebraically identical closed form the device path uses (the positive part of s u v^T is
u+ v+^T + u- v-^T with disjoint supports), kept here so the tests can check the two against each
other in float64.  Pinned against goldens generated from the real reference
(tests/golden/gen_golden.py, cases nndsvd_*).
"""
import numpy as np
from .nmf_oracle import NMFOracle

_SVD_EPS = 10 ** -8                                   # svd.py:74


def dense_svd(data):
    """pymf.SVD(data).factorize() for dense data: returns U, S (diagonal matrix), V with data = U S V."""
    rows, cols = data.shape
    if rows > cols:                                   # svd.py:237-241 -> _left_svd (svd.py:127-148)
        AA = np.dot(data[:, :].T, data[:, :])
        values, v_vectors = np.linalg.eigh(AA)
        v_vectors = v_vectors[:, values > _SVD_EPS]
        values = values[values > _SVD_EPS]
        idx = np.argsort(values)[::-1]
        values = values[idx]
        S = np.diag(np.sqrt(values))
        S_inv = np.diag(1.0 / np.sqrt(values))
        Vtmp = v_vectors[:, idx]
        U = np.dot(np.dot(data[:, :], Vtmp), S_inv)
        V = Vtmp.T
    else:                                             # svd.py:242-246 -> _right_svd (svd.py:106-124)
        AA = np.dot(data[:, :], data[:, :].T)
        values, u_vectors = np.linalg.eigh(AA)
        u_vectors = u_vectors[:, values > _SVD_EPS]
        values = values[values > _SVD_EPS]
        idx = np.argsort(values)
        values = values[idx[::-1]]
        U = u_vectors[:, idx[::-1]]
        S = np.diag(np.sqrt(values))
        S_inv = np.diag(np.sqrt(values) ** -1)
        V = np.dot(S_inv, np.dot(U[:, :].T, data[:, :]))
    return U, S, V


class NNDSVDOracle(NMFOracle):
    def init_w(self):                                 # nndsvd.py:69-70
        self.W = np.zeros((self._data_dimension, self._num_bases))

    def init_h(self):                                 # nndsvd.py:72-73
        self.H = np.zeros((self._num_bases, self._num_samples))

    def update_h(self):                               # nndsvd.py:75-76
        pass

    def update_w(self):                               # nndsvd.py:78-106
        U, S, V = dense_svd(self.data)
        self.W[:, 0] = np.sqrt(S[0, 0]) * np.abs(U[:, 0])
        self.H[0, :] = np.sqrt(S[0, 0]) * np.abs(V[0, :].T)
        for i in range(1, self._num_bases):
            Tmp = np.dot(U[:, i:i + 1] * S[i, i], V[i:i + 1, :])
            Tmp = np.where(Tmp < 0, 0.0, Tmp)
            u, s, v = dense_svd(Tmp)
            self.W[:, i] = np.sqrt(s[0, 0]) * np.abs(u[:, 0])
            self.H[i, :] = np.sqrt(s[0, 0]) * np.abs(v[0, :].T)

    def factorize(self, niter=1, show_progress=False,
                  compute_w=True, compute_h=True, compute_err=True):   # nndsvd.py:108-114
        NMFOracle.factorize(self, niter=1, compute_w=True, compute_h=True, compute_err=compute_err)


def nndsvd_closed_form(data, k):
    """W, H of NNDSVD in float64 without the second SVDs (what the HIP path evaluates)."""
    A = np.asarray(data, dtype=np.float64)
    U, S, V = dense_svd(A)
    m, n = A.shape
    W = np.zeros((m, k))
    H = np.zeros((k, n))
    s = np.diag(S)
    W[:, 0] = np.sqrt(s[0]) * np.abs(U[:, 0])
    H[0, :] = np.sqrt(s[0]) * np.abs(V[0, :])
    for i in range(1, k):
        u, v = U[:, i], V[i, :]
        up, un = np.maximum(u, 0), np.maximum(-u, 0)
        vp, vn = np.maximum(v, 0), np.maximum(-v, 0)
        a = np.linalg.norm(up) * np.linalg.norm(vp)
        b = np.linalg.norm(un) * np.linalg.norm(vn)
        if a >= b:
            sig = s[i] * a
            W[:, i] = np.sqrt(sig) * up / np.linalg.norm(up)
            H[i, :] = np.sqrt(sig) * vp / np.linalg.norm(vp)
        else:
            sig = s[i] * b
            W[:, i] = np.sqrt(sig) * un / np.linalg.norm(un)
            H[i, :] = np.sqrt(sig) * vn / np.linalg.norm(vn)
    return W, H
