"""pymf_amd.NMFNNLS -- drop-in for pymf.NMFNNLS (reference pymf/nmfnnls.py) on MI355X.

The reference solves, per column of H / row of W, `scipy.optimize.nnls(W, data[:, i])`
(nmfnnls.py:69-80): argmin ||W x - v||, x >= 0.  That is the same strictly convex
problem as NMFALS' QP (1/2 x'(W'W)x - (W'v)'x, x >= 0, nmfals.py:70-97), so both
classes share the batched exact active-set kernel `k_nnqp`.
"""
from .nmfals import NMFALS

__all__ = ["NMFNNLS"]


class NMFNNLS(NMFALS):
    _SHIPPED = True
