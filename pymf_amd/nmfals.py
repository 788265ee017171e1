"""pymf_amd.NMFALS -- drop-in for pymf.NMFALS (reference pymf/nmfals.py) on MI355X.

Alternating least squares: each half step solves, for every row of W (resp.
column of H), the QP  min 1/2 x'HA x + FA'x  s.t. x >= 0  with the shared
Hessian HA = H H^T (resp. W^T W) and FA = -H v^T (resp. -W^T v)
(nmfals.py:70-97).  The reference calls cvxopt.solvers.qp once per sub-problem;
here the Gram/right-hand sides are MFMA contractions and all sub-problems are
solved at once by a batched exact active-set kernel (one wave per problem).
"""
from . import _lib
from .nmf import NMF

__all__ = ["NMFALS"]


class NMFALS(NMF):
    _SHIPPED = True
    _ALGO = _lib.ALGO_NMFALS
