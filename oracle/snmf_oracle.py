"""NumPy restatement of pymf.SNMF (reference pymf/snmf.py:67-91) -- test oracle."""
import numpy as np
from .nmf_oracle import NMFOracle, EPS_DEN


def snmf_update_w(V, W, H):
    """snmf.py:67-70 -- W = (V H^T) inv(H H^T); returns a NEW array (rebinding)."""
    W1 = np.dot(V, H.T)                     # :68
    W2 = np.dot(H, H.T)                     # :69
    return np.dot(W1, np.linalg.inv(W2))    # :70


def snmf_update_h(V, W, H):
    """snmf.py:72-91 -- H *= sqrt((XW+ + H^T WW-)^T / ((XW- + H^T WW+)^T + 1e-9))."""
    def pos(m):
        return (np.abs(m) + m) / 2.0        # :73-74

    def neg(m):
        return (np.abs(m) - m) / 2.0        # :76-77

    XW = np.dot(V.T, W)                     # :79
    WW = np.dot(W.T, W)                     # :81
    WW_pos = pos(WW)                        # :82
    WW_neg = neg(WW)                        # :83
    XW_pos = pos(XW)                        # :85
    H1 = (XW_pos + np.dot(H.T, WW_neg)).T   # :86
    XW_neg = neg(XW)                        # :88
    H2 = (XW_neg + np.dot(H.T, WW_pos)).T + EPS_DEN   # :89
    H *= np.sqrt(H1 / H2)                   # :91
    return H


class SNMFOracle(NMFOracle):
    update_w_fn = staticmethod(snmf_update_w)
    update_h_fn = staticmethod(snmf_update_h)
    rebinding_w = True                      # snmf.py:70 rebinds self.W
