"""Restatement of pymf.NMFALS (reference pymf/nmfals.py:70-97) -- test oracle.

The reference hands every column/row sub-problem to cvxopt.solvers.qp
(third-party, un-pinned in setup.py:12-16, not installed here):

    minimise 1/2 x^T HA x + FA^T x   subject to  -I x <= 0          (nmfals.py:74,89)

with HA = W^T W (resp. H H^T) and FA = -W^T v (resp. -H v^T), all float64
(nmfals.py:73,78,88,93).  HA is a Gram matrix, positive definite whenever the
factor has full column rank, so the minimiser is unique; this oracle computes
it exactly with a float64 active-set (Lawson-Hanson on the normal equations).
cvxopt's interior-point iterate differs from it by its stopping tolerance
(abstol 1e-7 / reltol 1e-6), hence "parity unpinned" at digit level; tests use
abs 1e-6 and pin the oracle against the reference's NNLS sibling.
"""
import numpy as np
from .nmf_oracle import NMFOracle


def nnqp_solve(HA, FA, max_outer=None):
    """argmin_x 1/2 x'HA x + FA'x, x >= 0, for ONE problem (float64, exact active set).

    Lawson-Hanson NNLS expressed on the Gram matrix (HA = A'A, -FA = A'b).
    """
    HA = np.atleast_2d(np.asarray(HA, dtype=np.float64))   # np.float64(1 x 1 array) is a scalar (num_bases = 1)
    f = -np.asarray(FA, dtype=np.float64).ravel()      # A'b
    k = f.shape[0]
    x = np.zeros(k)
    passive = np.zeros(k, dtype=bool)
    w = f.copy()                                       # negative gradient at x = 0
    tol = 10 * np.finfo(np.float64).eps * np.abs(HA).sum(axis=0).max() * k
    max_outer = max_outer or 30 * k
    for _ in range(max_outer):
        if passive.all() or not (w[~passive] > tol).any():
            break
        passive[int(np.argmax(np.where(~passive, w, -np.inf)))] = True
        while True:
            idx = np.flatnonzero(passive)
            s = np.zeros(k)
            s[idx] = np.linalg.solve(HA[np.ix_(idx, idx)], f[idx])
            if (s[idx] > 0).all():
                break
            bad = idx[s[idx] <= 0]
            alpha = np.min(x[bad] / (x[bad] - s[bad]))
            x = x + alpha * (s - x)
            passive[idx[x[idx] <= 1e-15 * max(1.0, np.abs(x).max())]] = False
            x[~passive] = 0.0
        x = s
        w = f - HA.dot(x)
    return x


def als_update_h(V, W, H):
    """nmfals.py:70-82 -- every column of H solves the QP with HA = W^T W."""
    HA = np.float64(np.dot(W.T, W))                    # :78
    for i in range(V.shape[1]):                        # :82 (eager map in Py2)
        FA = np.float64(np.dot(-W.T, V[:, i]))         # :73
        H[:, i] = nnqp_solve(HA, FA)                   # :74-75
    return H


def als_update_w(V, W, H):
    """nmfals.py:85-97 -- every row of W solves the QP with HA = H H^T."""
    HA = np.float64(np.dot(H, H.T))                    # :93
    for i in range(V.shape[0]):                        # :97
        FA = np.float64(np.dot(-H, V[i, :].T))         # :88
        W[i, :] = nnqp_solve(HA, FA)                   # :89-90
    return W


class NMFALSOracle(NMFOracle):
    update_w_fn = staticmethod(als_update_w)
    update_h_fn = staticmethod(als_update_h)
