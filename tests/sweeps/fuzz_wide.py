#!/usr/bin/env python3
"""Randomised sweep of SNMF over every inverse kernel (k_inverse_spd_mfma<4>, <8>, k_inverse_spd_big) and of the
wide-base generic paths (NMFALS / NMFNNLS > 64, SNMF / RNMF > 128) against the float64 oracles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymf_amd
import oracle
from pymf_amd.rnmf import RNMF

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rel = lambda a, b: np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-30)
bad = 0
for t in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    cls = str(rs.choice(["SNMF", "SNMF", "SNMF", "NMFALS", "RNMF"]))
    m = int(rs.choice([70, 300, 1000, 2500])); n = int(rs.choice([40, 64, 130, 256, 400, 700]))
    if cls == "SNMF":
        k = int(rs.choice([1, 3, 15, 16, 17, 31, 33, 48, 63, 64, 65, 80, 100, 127, 128, 129, 200, 300]))
        k = min(k, max(1, int(0.8 * n)))        # k close to n: H H^T ill-conditioned, a different experiment
    elif cls == "NMFALS":
        k = int(rs.choice([65, 70, 100, 128, 129, 160])); k = min(k, max(2, n // 2), m // 2)
    else:
        k = int(rs.choice([129, 150, 260])); k = min(k, n, m)
    niter = int(rs.choice([1, 2, 4])); hooks = bool(rs.random_sample() < 0.3)
    try:
        if cls == "RNMF":
            V = rs.random_sample((m, n)).astype(np.float32)
            V.flat[rs.randint(0, V.size, size=max(1, V.size // 300))] += 5.0
            seed = int(rs.randint(1 << 30))
            np.random.seed(seed); a = RNMF(V, num_bases=k, lamb=1.0); a.factorize(niter=niter)
            np.random.seed(seed); o = oracle.RNMFOracle(V, num_bases=k, lamb=1.0); o.factorize(niter=niter)
        else:
            V = (rs.random_sample((m, n)) - (0.4 if cls == "SNMF" else 0.0)).astype(np.float32)
            W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
            a = getattr(pymf_amd, cls)(V, num_bases=k); a.W, a.H = W0.copy(), H0.copy()
            o = getattr(oracle, cls + "Oracle")(V.astype(np.float64), num_bases=k); o.W, o.H = W0.copy(), H0.copy()
            if hooks:
                for _ in range(niter):
                    a.update_w(); o.update_w(); a.update_h(); o.update_h()
            else:
                a.factorize(niter=niter); o.factorize(niter=niter)
        e = max(rel(a.W, o.W), rel(a.H, o.H))
        tol = 5e-4 if cls == "NMFALS" else 1e-4
        flag = "" if e < tol else "  <<<<<"
        bad += bool(flag)
        print(cls, m, n, k, niter, "hooks" if hooks else "loop", "rel %.1e" % e, flag, flush=True)
    except Exception as ex:
        bad += 1; print(cls, m, n, k, "EXC", type(ex).__name__, str(ex)[:120], flush=True)
print("bad", bad)
