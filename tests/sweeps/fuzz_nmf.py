#!/usr/bin/env python3
"""Randomised odd-input sweep of the pymf_amd classes against the oracle (dtype, layout, m/n/k of 1,
k > n, every fused / split / tiled shape class).  ferr is compared relative to ||V||: an (almost)
exact fit leaves a float32-sized residual floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymf_amd
from pymf_amd.rnmf import RNMF
from oracle import NMFOracle, SNMFOracle, BNMFOracle, RNMFOracle

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ntrial = int(sys.argv[2]) if len(sys.argv) > 2 else 60


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


bad = 0
for trial in range(ntrial):
    m = int(rs.choice([1, 2, 3, 17, 64, 65, 200, 1000, 5000]))
    n = int(rs.choice([1, 2, 5, 63, 64, 65, 130, 190, 257, 320, 384, 400, 450, 512, 600]))
    k = int(rs.choice([1, 2, 3, 16, 17, 32, 33, 64, 65, 100, 128]))
    kind = rs.choice(["f32", "f64", "fortran", "slice", "int"])
    name = rs.choice(["NMF", "NMF", "SNMF", "BNMF", "RNMF"])
    V = rs.random_sample((m, n))
    if name == "BNMF":
        V = (V < 0.3).astype(np.float64)
    if name == "SNMF" and 2 * k > min(m, n):
        name = "NMF"                      # H H^T singular: inv() is noise in the reference as well
    if name == "NMF" and rs.random_sample() < 0.15:
        k = int(rs.choice([129, 200, 256, 300]))      # blocks of 128 bases
    if kind == "f32": V = V.astype(np.float32)
    elif kind == "fortran": V = np.asfortranarray(V.astype(np.float32))
    elif kind == "slice": V = np.ascontiguousarray(np.tile(V, (2, 2)).astype(np.float32))[::2, ::2]
    elif kind == "int": V = (V * 10).astype(np.int64)
    Vo = np.asarray(V, dtype=np.float64) if kind == "int" else V
    W0 = rs.random_sample((m, k)); H0 = rs.random_sample((k, n))
    try:
        if name == "RNMF":
            fl = dict(compute_w=bool(rs.random_sample() < 0.8), compute_h=bool(rs.random_sample() < 0.8))
            np.random.seed(trial); a = RNMF(V, num_bases=k, lamb=0.7); a.factorize(niter=3); a.factorize(niter=4, **fl)
            np.random.seed(trial); o = RNMFOracle(Vo, num_bases=k, lamb=0.7); o.factorize(niter=3); o.factorize(niter=4, **fl)
            tolw, tolf = 5e-3, 5e-4          # the soft threshold is discontinuous at float32 rounding
        else:
            cls = getattr(pymf_amd, name); orc = {"NMF": NMFOracle, "SNMF": SNMFOracle, "BNMF": BNMFOracle}[name]
            a = cls(V, num_bases=k); a.W, a.H = W0.copy(), H0.copy()
            o = orc(Vo, num_bases=k); o.W, o.H = W0.copy(), H0.copy()
            # two calls with random niter / flags (the reference's implicit resume)
            for call in range(2):
                niter = int(rs.randint(1, 13))
                flags = dict(compute_w=bool(rs.random_sample() < 0.8), compute_h=bool(rs.random_sample() < 0.8),
                             compute_err=bool(call == 1 or rs.random_sample() < 0.7))
                a.factorize(niter=niter, **flags)
                o.factorize(niter=niter, **flags)
                if flags["compute_err"] and len(a.ferr) != len(o.ferr):
                    # nmf.py:134-139 stops on |ferr[i] - ferr[i-1]| / n < 1e-8.  The device path carries float32-sized
                    # noise in ferr (W, H are stored in float32): where that noise exceeds the threshold the decision
                    # cannot be reproduced digit for digit and a different stopping iteration is a NOTE; where the
                    # threshold is well above the noise a difference is a FAILURE.
                    # (an exact fit -- k >= n, rank-deficient data -- leaves a residual of float32 rounding of W, H: its size is
                    # eps32 ||V||, whatever ferr itself has shrunk to)
                    scale = max(float(np.max(np.abs(a.ferr))) if len(a.ferr) else 0.0, float(np.max(np.abs(o.ferr))) if len(o.ferr) else 0.0,
                                float(np.linalg.norm(np.asarray(Vo, dtype=np.float64))))
                    noise = 16.0 * float(np.finfo(np.float32).eps) * scale
                    if 1e-8 * n > noise:
                        bad += 1
                        print(m, n, k, kind, name, "FAIL: stopped after %d vs %d iterations with the threshold %.1e above the float32 noise %.1e  <<<<<"
                              % (len(a.ferr), len(o.ferr), 1e-8 * n, noise))
                    else:
                        print(m, n, k, kind, name, "note: stopped after %d vs %d iterations (threshold %.1e inside the float32 noise %.1e of ferr)"
                              % (len(a.ferr), len(o.ferr), 1e-8 * n, noise))
                    o.W, o.H = a.W.copy().astype(o.W.dtype), a.H.copy().astype(o.H.dtype)
                    o.ferr = np.asarray(a.ferr).copy()
            tolw, tolf = (2e-5, 2e-6) if name != "SNMF" else (2e-3, 2e-5)
        if not (np.isfinite(o.W).all() and np.isfinite(o.H).all()) or np.linalg.norm(Vo) == 0:
            print(m, n, k, kind, name, "skipped: the reference itself divides by zero here (no epsilon) / V = 0")
            continue
        e = max(rel(a.W, o.W), rel(a.H, o.H))
        fe = abs(a.ferr[-1] - o.ferr[-1]) / max(np.linalg.norm(np.asarray(V, dtype=np.float64)), 1e-12)
        flag = "" if (e < tolw and fe < tolf) else "  <<<<<"
        if flag: bad += 1
        print(m, n, k, kind, name, getattr(a._ctx, "path_name", "?"), "relWH %.2e ferr %.2e" % (e, fe), flag)
    except Exception as ex:
        bad += 1; print(m, n, k, kind, name, "EXC", type(ex).__name__, str(ex)[:120])
print("bad", bad)
