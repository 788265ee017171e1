#!/usr/bin/env python3
"""ms of each of the first NMFALS iterations from the random start at cfg3's shape, `nnqp_frame16` on and off."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymf_amd import _lib
m, n, k = 262144, 1024, 64
for f16 in (1, 0):
    ctx = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
    ctx.set_option("nnqp_frame16", f16)
    ctx.fill_v_uniform(1234); ctx.fill_w_uniform(42); ctx.fill_h_uniform(43)
    ts = []
    for it in range(16):
        ctx.factorize(1, compute_err=False); ctx.synchronize()
        ts.append(ctx.last_loop_ms())
    print("frame16=%d:" % f16, " ".join("%.2f" % t for t in ts))
    ctx.close()
