#!/usr/bin/env python3
"""NNDSVD through the two eigen-solvers of pmf_nndsvd_init (full Jacobi / filtered top-k subspace iteration) on a few
spectra: agreement of the two, distance from the float64 oracle, wall time.  PMF_TOPK_DEBUG=1 prints the iterations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pymf_amd import _lib
from oracle import nndsvd_closed_form


def rel(a, b): return np.linalg.norm(a - b) / np.linalg.norm(b)


def make(kind, m, n, rs):
    if kind == "uniform": return rs.random_sample((m, n))
    if kind == "lowrank": return rs.random_sample((m, 40)) @ rs.random_sample((40, n)) + 0.01 * rs.random_sample((m, n))
    if kind == "binary": return (rs.random_sample((m, n)) < 0.05).astype(np.float64)
    if kind == "dupcols":
        a = rs.random_sample((m, n // 2)); return np.concatenate([a, a], axis=1)
    if kind == "diagish":
        a = 0.01 * rs.random_sample((m, n)); a[np.arange(m) % n == np.arange(m)[:, None] % n if False else (np.arange(m)[:, None] % n == np.arange(n)[None, :])] += 1.0; return a
    raise ValueError(kind)


cases = [("uniform", 3000, 1500, 12), ("uniform", 6000, 2048, 100), ("lowrank", 4000, 1536, 64), ("binary", 5000, 2048, 48),
         ("dupcols", 3000, 1024, 20), ("diagish", 4096, 1024, 32), ("uniform", 5000, 4608, 32), ("uniform", 9000, 8192, 64),
         ("uniform", 4096, 2048, 256), ("uniform", 17000, 16384, 64), ("uniform", 12000, 8000, 512)]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:] or str(c[2]) in sys.argv[1:]]
for (kind, m, n, k) in cases:
    V = make(kind, m, n, np.random.RandomState(m + n + k)).astype(np.float32)
    res = {}
    for topk in ((1, 0) if n <= 4096 else (1,)):
        ctx = _lib.Context(_lib.ALGO_NMF, m, n, k)
        ctx.set_v_dense(V)
        ctx.set_option("nndsvd_topk", topk)
        t = time.time()
        try:
            found = ctx.nndsvd_init()
        except _lib.PmfError as e:
            print("%s m %d n %d k %d topk %d: %s" % (kind, m, n, k, topk, str(e)[:150]), flush=True)
            ctx.close()
            continue
        dt = time.time() - t
        res[topk] = (ctx.get_w(), ctx.get_h())
        print("%s m %d n %d k %d topk %d: found %d in %.2f s" % (kind, m, n, k, topk, found, dt), flush=True)
        ctx.close()
    if 0 in res and 1 in res:
        print("   top-k vs Jacobi: W %.2e H %.2e" % (rel(res[1][0], res[0][0]), rel(res[1][1], res[0][1])))
    if 1 in res and n <= 5000:
        t = time.time(); Wr, Hr = nndsvd_closed_form(V, k)
        print("   top-k vs float64 oracle (%.1f s): W %.2e H %.2e" % (time.time() - t, rel(res[1][0], Wr), rel(res[1][1], Hr)), flush=True)
