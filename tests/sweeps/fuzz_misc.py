#!/usr/bin/env python3
"""Randomised sweeps of the streamed NMF path (vs the resident one) and of NNDSVD (vs the float64 closed form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymf_amd
from oracle import nndsvd_closed_form

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
bad = 0
for t in range(40):
    m = int(rs.choice([1, 3, 64, 65, 130, 1000, 5000])); n = int(rs.choice([1, 5, 64, 100, 257, 400, 600]))
    k = int(rs.choice([1, 4, 16, 33, 64, 100, 128])); rows = int(rs.choice([64, 128, 512, 4096]))
    V = rs.random_sample((m, n)).astype(np.float32)
    W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
    fl = dict(compute_w=bool(rs.random_sample() < 0.8), compute_h=bool(rs.random_sample() < 0.8))
    try:
        a = pymf_amd.NMF(V, num_bases=k); a.W, a.H = W0.copy(), H0.copy(); a.factorize(niter=4, **fl)
        b = pymf_amd.NMF(V, num_bases=k); b.stream_rows = rows; b.W, b.H = W0.copy(), H0.copy(); b.factorize(niter=4, **fl)
        e = max(rel(b.W, a.W), rel(b.H, a.H)); fe = abs(a.ferr[-1] - b.ferr[-1]) / max(np.linalg.norm(V), 1e-12) if len(a.ferr) == len(b.ferr) else -1
        if fe < 0:       # the two paths stopped at different iterations: a convergence decision at float32 resolution
            print("stream", m, n, k, rows, fl, "note: stopped after %d vs %d iterations" % (len(a.ferr), len(b.ferr)))
            continue
        flag = "" if (e < 1e-5 and 0 <= fe < 2e-6) else "  <<<<<"
        bad += bool(flag); print("stream", m, n, k, rows, fl, "relWH %.1e ferr %.1e" % (e, fe), len(a.ferr), len(b.ferr), flag)
    except Exception as ex:
        bad += 1; print("stream", m, n, k, rows, "EXC", type(ex).__name__, str(ex)[:100])
for t in range(30):
    m = int(rs.choice([2, 30, 64, 200, 1000, 3000])); n = int(rs.choice([2, 17, 64, 130, 300, 900]))
    k = int(rs.randint(1, min(m, n, 64) + 1))
    V = rs.random_sample((m, n)).astype(np.float32) + (0.5 if rs.random_sample() < 0.5 else 0.0)
    try:
        a = pymf_amd.NNDSVD(V, num_bases=k); a.factorize()
        W, H = nndsvd_closed_form(V, k)
        # trailing singular directions of random data are close to degenerate: compare the leading half tightly
        kk = max(1, k // 2)
        e = max(rel(a.W[:, :kk], W[:, :kk]), rel(a.H[:kk], H[:kk]))
        flag = "" if e < 5e-3 else "  <<<<<"
        bad += bool(flag); print("nndsvd", m, n, k, "rel(leading %d) %.1e all %.1e" % (kk, e, max(rel(a.W, W), rel(a.H, H))), flag)
    except Exception as ex:
        bad += 1; print("nndsvd", m, n, k, "EXC", type(ex).__name__, str(ex)[:100])
print("bad", bad)
