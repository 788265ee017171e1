#!/usr/bin/env python3
"""More shapes at the edges of the index ranges (device-filled data, checks on the host in float64):
  1. NMFALS on a V beyond 2^32 elements: KKT conditions of sampled row QPs over the whole row range, monotone objective;
  2. SNMF on the same V: sampled rows of W = V M^T against float64 (M^T = inv(H H^T) H), H step finite;
  3. NMF on a WIDE matrix (few rows, 10^6 columns): against the float64 oracle.
python3 tests/sweeps/huge_probe2.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pymf_amd import _lib
import oracle
from test_gpu_parity import _synthetic_rows

m = int(sys.argv[1]) if len(sys.argv) > 1 else 20 * 1024 * 1024
n = 256
bad = 0
rows = np.unique(np.concatenate([np.arange(0, m, 300007), np.arange(16777210, 16777222) % m, np.arange(m - 40, m)]))

# ---- 1. NMFALS, k = 16 ----
t0 = time.time()
k = 16
c = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
c.fill_v_uniform(1234); c.fill_w_uniform(42); c.fill_h_uniform(43)
f0 = c.frobenius()
H0 = c.get_h().astype(np.float64)
c.update_w()
W1 = c.get_w()
Vs = _synthetic_rows(1234, rows, n).astype(np.float64)
HA, Fm = H0.dot(H0.T), Vs.dot(H0.T)
Ws = W1[rows].astype(np.float64)
g = Ws.dot(HA) - Fm
scale = np.abs(Fm).max()
ok = (W1.min() >= 0.0) and np.isfinite(W1).all() and g.min() > -2e-4 * scale and np.abs(Ws * g).max() < 2e-4 * scale * max(1.0, Ws.max())
f1 = c.frobenius(); c.update_h(); f2 = c.frobenius()
ok = ok and f1 <= f0 * (1 + 1e-6) and f2 <= f1 * (1 + 1e-6)
print("NMFALS %d x %d (%.1f G elements), k = %d: KKT of %d sampled rows: min gradient %.2e, complementarity %.2e (scale %.1f); ferr %.4f -> %.4f -> %.4f; %s; %.1f s"
      % (m, n, m * n / 2**30, k, len(rows), g.min(), np.abs(Ws * g).max(), scale, f0, f1, f2, "ok" if ok else "BAD", time.time() - t0), flush=True)
bad += not ok
del W1
c.close()

# ---- 2. SNMF, k = 32 ----
t0 = time.time()
k = 32
c = _lib.Context(_lib.ALGO_SNMF, m, n, k)
c.fill_v_uniform(1234); c.fill_w_uniform(42); c.fill_h_uniform(43)
H0 = c.get_h().astype(np.float64)
c.update_w()
W1 = c.get_w()
Wref = Vs.dot(H0.T).dot(np.linalg.inv(H0.dot(H0.T)))               # snmf.py:67-70
e = np.linalg.norm(W1[rows] - Wref) / np.linalg.norm(Wref)
c.update_h()
H1 = c.get_h()
ok = e < 2e-5 and np.isfinite(H1).all()
print("SNMF   %d x %d, k = %d: %d sampled rows of W = V H^T inv(H H^T) vs float64: rel %.2e; H finite %s; %s; %.1f s"
      % (m, n, k, len(rows), e, bool(np.isfinite(H1).all()), "ok" if ok else "BAD", time.time() - t0), flush=True)
bad += not ok
del W1
c.close()

# ---- 3. NMF, wide: 1024 x 1 000 000, k = 16 ----
t0 = time.time()
mw, nw, k = 1024, 1000000, 16
c = _lib.Context(_lib.ALGO_NMF, mw, nw, k)
c.fill_v_uniform(7); c.fill_w_uniform(8); c.fill_h_uniform(9)
V = _synthetic_rows(7, np.arange(mw), nw)
o = oracle.NMFOracle(V, num_bases=k)
o.W, o.H = c.get_w().astype(np.float64), c.get_h().astype(np.float64)
fe, done, conv = c.factorize(2, compute_err=True)
o.factorize(niter=2)
eW = np.linalg.norm(c.get_w() - o.W) / np.linalg.norm(o.W)
eH = np.linalg.norm(c.get_h() - o.H) / np.linalg.norm(o.H)
ef = np.max(np.abs(np.asarray(fe)[:done] - o.ferr) / o.ferr)
ok = eW < 2e-5 and eH < 2e-5 and ef < 1e-5
print("NMF    %d x %d, k = %d (path %s): relW %.2e relH %.2e ferr rel %.2e; %s; %.1f s" % (mw, nw, k, c.path_name, eW, eH, ef, "ok" if ok else "BAD", time.time() - t0), flush=True)
bad += not ok
c.close()
print("bad %d" % bad)
