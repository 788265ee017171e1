#!/usr/bin/env python3
"""ms of the first NMFALS iterations beyond 64 bases (k_nnqp_big / its successors): argv = m n k [iters]."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymf_amd import _lib
m, n, k = (int(a) for a in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ctx = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
for a in sys.argv[5:]:
    name, val = a.split("=")
    ctx.set_option(name, int(val))
ctx.fill_v_uniform(1234); ctx.fill_w_uniform(42); ctx.fill_h_uniform(43)
ts = []
for it in range(iters):
    ctx.factorize(1, compute_err=False); ctx.synchronize()
    ts.append(ctx.last_loop_ms())
print("m=%d n=%d k=%d:" % (m, n, k), " ".join("%.1f" % t for t in ts))
ctx.factorize(1, compute_err=True)
print("ferr", ctx.ferr()[-1] if hasattr(ctx, "ferr") else None)
ctx.close()
