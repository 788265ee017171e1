#!/usr/bin/env python3
"""Randomised sweep of NMFALS / NMFNNLS (exact active-set QP on the device) against the float64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymf_amd
from oracle import NMFALSOracle
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"      # 65 .. 128 bases: k_nnqp_wave
bad = 0
for t in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    m = int(rs.choice([3, 10, 40, 90])); n = int(rs.choice([3, 12, 50, 80])); k = int(rs.choice([1, 2, 5, 16, 17, 33, 64]))
    if wide:
        m = int(rs.choice([130, 200, 331])); n = int(rs.choice([129, 160, 260])); k = int(rs.choice([65, 72, 96, 100, 127, 128]))
    if k > min(m, n): k = min(m, n)          # Gram matrices of full rank: unique minimisers
    V = rs.random_sample((m, n)).astype(np.float32)
    if rs.random_sample() < 0.3: V[rs.random_sample((m, n)) < 0.6] = 0
    W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
    cls = pymf_amd.NMFALS if rs.random_sample() < 0.5 else pymf_amd.NMFNNLS
    niter = int(rs.randint(1, 6))
    try:
        a = cls(V, num_bases=k); a.W, a.H = W0.copy(), H0.copy(); a.factorize(niter=niter)
        o = NMFALSOracle(V, num_bases=k); o.W, o.H = W0.copy(), H0.copy(); o.factorize(niter=niter)
        if max(np.linalg.cond(o.W.T @ o.W), np.linalg.cond(o.H @ o.H.T)) > 1e8:
            print(m, n, k, cls.__name__, niter, "skipped: a Gram matrix is singular, the QP minimisers are not unique")
            continue
        ew = np.abs(a.W - o.W).max() / max(1.0, np.abs(o.W).max()); eh = np.abs(a.H - o.H).max() / max(1.0, np.abs(o.H).max())
        rec = np.linalg.norm(a.W @ a.H - o.W @ o.H) / max(np.linalg.norm(o.W @ o.H), 1e-30)
        flag = "" if (rec < 1e-4 and (max(ew, eh) < 2e-3)) else "  <<<<<"
        bad += bool(flag); print(m, n, k, cls.__name__, niter, "maxW %.1e maxH %.1e recon %.1e" % (ew, eh, rec), flag)
    except Exception as ex:
        bad += 1; print(m, n, k, "EXC", type(ex).__name__, str(ex)[:100])
print("bad", bad)
