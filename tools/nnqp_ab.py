#!/usr/bin/env python3
"""NMFALS half steps on k_nnqp_quad (16 lanes per problem) vs k_nnqp (lane = variable): same minimisers, timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymf_amd import _lib
m, n, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 1024, 64)
niter = int(sys.argv[4]) if len(sys.argv) > 4 else 4
res = {}
for quad in (2, 0):
    c = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
    c.set_option("nnqp_quad", quad)
    c.fill_v_uniform(1234); c.fill_w_uniform(42); c.fill_h_uniform(43)
    c.factorize(1, compute_err=False)
    t = time.time(); ferr, done, _ = c.factorize(niter, compute_err=True); dt = time.time() - t
    res[quad] = (c.get_w(), c.get_h(), ferr)
    print("nnqp_quad=%d: %.3f ms/iter, ferr %s" % (quad, dt / niter * 1e3, np.array2string(ferr, precision=6)))
    c.close()
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
print("W rel diff %.2e  H rel diff %.2e  max|dW| %.2e" % (rel(res[2][0], res[0][0]), rel(res[2][1], res[0][1]), np.abs(res[2][0] - res[0][0]).max()))
print("support mismatch (W): %d of %d entries" % (int(((res[2][0] > 0) != (res[0][0] > 0)).sum()), res[0][0].size))
