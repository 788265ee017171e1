#!/usr/bin/env python3
"""Wide-base path (num_bases > 128) on a matrix whose W alone has more than 2^32 elements (default 20 Mi rows x 256, k = 256):
sampled rows of W after update_w against the oracle's rule; the error of one iteration by the trace identity (factorize) against
the direct residual pass (hooks) -- two computations over all rows.   python3 tests/sweeps/huge_probe3.py [rows] [k]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pymf_amd import _lib
import oracle
from test_gpu_parity import _synthetic_rows

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20 * 1024 * 1024
k = int(sys.argv[2]) if len(sys.argv) > 2 else 256
n = 256
bad = 0
t0 = time.time()
rows = np.unique(np.concatenate([np.arange(0, m, 500009), np.arange(16777210, 16777222) % m, np.arange(m - 40, m)]))
a = _lib.Context(_lib.ALGO_NMF, m, n, k)
a.fill_v_uniform(1234); a.fill_w_uniform(42); a.fill_h_uniform(43)
H0 = a.get_h().astype(np.float64)
W0s = a.get_w()[rows].astype(np.float64)
a.update_w()
W1s = a.get_w()[rows].astype(np.float64)
Wref = W0s.copy()
oracle.nmf_update_w(_synthetic_rows(1234, rows, n), Wref, H0.copy())
e = np.linalg.norm(W1s - Wref) / np.linalg.norm(Wref)
per = np.linalg.norm(W1s - Wref, axis=1) / np.linalg.norm(Wref, axis=1)
print("NMF %d x %d, k = %d (W: %.1f G elements): %d sampled rows of W after update_w vs the oracle: rel %.2e, rows off %d"
      % (m, n, k, m * k / 2**30, len(rows), e, int((per > 1e-5).sum())), flush=True)
bad += not (e < 5e-6)
a.update_h()
f_direct = a.frobenius()
a.close()
b = _lib.Context(_lib.ALGO_NMF, m, n, k)
b.fill_v_uniform(1234); b.fill_w_uniform(42); b.fill_h_uniform(43)
fe, done, conv = b.factorize(1, compute_err=True)
print("error after one iteration: direct residual pass (hooks) %.6f, factorize() %.6f: rel %.2e; %.1f s" % (f_direct, fe[0], abs(fe[0] - f_direct) / f_direct, time.time() - t0))
bad += not (abs(fe[0] - f_direct) <= 2e-6 * f_direct)
b.close()
print("bad %d" % bad)
