#!/usr/bin/env python3
"""A V beyond 2^31 elements (default 36 Mi rows x 256 = 36 GiB, k = 64; W 9 GiB): sampled rows of W after one update_w against
the oracle's rule on the rebuilt rows (row-local), and the trace-identity error of the loop against the direct residual pass
(two different kernels over all rows).   python3 tests/sweeps/huge_probe.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pymf_amd import _lib
import oracle
from test_gpu_parity import _synthetic_rows

m = int(sys.argv[1]) if len(sys.argv) > 1 else 36 * 1024 * 1024
n, k = 256, 64
bad = 0
t0 = time.time()
a = _lib.Context(_lib.ALGO_NMF, m, n, k)
a.fill_v_uniform(1234); a.fill_w_uniform(42); a.fill_h_uniform(43)
H0 = a.get_h().astype(np.float64)
rows = np.unique(np.concatenate([np.arange(0, m, 400009), np.arange(8388600, 8388616), np.arange(16777210, 16777222) % m,
                                 np.arange(m - 70, m)]))
W0 = a.get_w()
W0s = W0[rows].astype(np.float64)
del W0
a.update_w()
W1 = a.get_w()
Wref = W0s.copy()
oracle.nmf_update_w(_synthetic_rows(1234, rows, n), Wref, H0.copy())
e = np.linalg.norm(W1[rows] - Wref) / np.linalg.norm(Wref)
per = np.linalg.norm(W1[rows] - Wref, axis=1) / np.linalg.norm(Wref, axis=1)
badrows = rows[per > 1e-5]
print("rows off: %d of %d; first %s last %s; smallest bad row %s" % (len(badrows), len(rows), badrows[:4], badrows[-3:], badrows.min() if len(badrows) else None))
print("%d x %d (%.1f G elements), k = %d: %d sampled rows of W after update_w vs the oracle: rel %.2e  (path %s)" % (m, n, m * n / 2**30, k, len(rows), e, a.path_name))
bad += not (e < 2e-6)
del W1
a.update_h()
fe, done, conv = a.factorize(3, compute_err=True)
H1 = a.get_h().astype(np.float64)
a.close()
# the same three + one iterations on the two-pass kernels (different kernels, every row of V enters H through W^T V)
b = _lib.Context(_lib.ALGO_NMF, m, n, k)
b.set_option("force_tiled", 1)
b.fill_v_uniform(1234); b.fill_w_uniform(42); b.fill_h_uniform(43)
b.update_w(); b.update_h()
fe2, done2, conv2 = b.factorize(3, compute_err=True)
H2 = b.get_h().astype(np.float64)
eh = np.linalg.norm(H1 - H2) / np.linalg.norm(H2)
ef = np.max(np.abs(np.asarray(fe)[:done] - np.asarray(fe2)[:done2]) / np.asarray(fe2)[:done2])
print("one-pass (%s) vs two-pass (%s) kernels after 4 iterations: H rel %.2e, ferr rel %.2e, ferr %s; %.1f s" % (a.path_name, b.path_name, eh, ef, np.asarray(fe)[:3], time.time() - t0))
bad += (not (eh < 2e-5)) + (not (ef < 1e-5)) + (not np.all(np.diff(np.asarray(fe)[:done]) <= 0))
b.close()
print("bad %d" % bad)
