#!/usr/bin/env python3
"""update_w against the float64 oracle as the number of columns grows (one accumulation chain of V H^T spans n / 4 fp32 MFMA
steps up to 65 536 columns, chunks of 65 536 beyond: PMF_WIDE_K in pmf_api.hip): mean / std / max relative error of W."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pymf_amd import _lib
import oracle
from test_gpu_parity import _synthetic_rows

bad = 0
SHAPES = ((_lib.ALGO_NMF, "NMF", 1024, 32768, 16), (_lib.ALGO_NMF, "NMF", 1024, 65536, 16), (_lib.ALGO_NMF, "NMF", 1024, 131072, 16),
          (_lib.ALGO_NMF, "NMF", 1024, 1000000, 16), (_lib.ALGO_NMF, "NMF", 256, 1000000, 64), (_lib.ALGO_NMF, "NMF", 300, 200000, 130))
if "--quick" in sys.argv:                   # (the GPU suite's sample)
    SHAPES = (SHAPES[0], SHAPES[2], (_lib.ALGO_NMF, "NMF", 512, 1000000, 16), SHAPES[5])
for (algo, name, mw, nw, k) in SHAPES:
    c = _lib.Context(algo, mw, nw, k)
    c.fill_v_uniform(7); c.fill_w_uniform(8); c.fill_h_uniform(9)
    V = _synthetic_rows(7, np.arange(mw), nw).astype(np.float64)
    W0, H0 = c.get_w().astype(np.float64), c.get_h().astype(np.float64)
    c.update_w()
    W1 = c.get_w().astype(np.float64)
    Wr = W0.copy(); oracle.nmf_update_w(V, Wr, H0.copy())
    d = (W1 - Wr) / np.maximum(np.abs(Wr), 1e-3 * np.abs(Wr).max())
    e = np.linalg.norm(W1 - Wr) / np.linalg.norm(Wr)
    ok = e < 2e-5
    print("%-6s n = %7d k = %3d path %-16s: update_w vs oracle: rel %.2e; element-wise mean %+.2e std %.2e max %.2e %s" %
          (name, nw, k, c.path_name, e, d.mean(), d.std(), np.abs(d).max(), "" if ok else "BAD"), flush=True)
    bad += not ok
    c.close()
# the other classes through their host classes: one iteration, W and H against the oracle classes
import pymf_amd
from pymf_amd.bnmf import BNMF
from pymf_amd.rnmf import RNMF
for (name, cls, ocls, mw, nw, k, kw) in (("BNMF", BNMF, oracle.BNMFOracle, 512, 300000, 32, {}), ("SNMF", pymf_amd.SNMF, oracle.SNMFOracle, 512, 300000, 32, {}),
                                         ("NMFALS", pymf_amd.NMFALS, oracle.NMFALSOracle, 400, 300000, 16, {}), ("RNMF", RNMF, oracle.RNMFOracle, 512, 300000, 16, {"lamb": 1.0})):
    rs = np.random.RandomState(nw + k)
    V = rs.random_sample((mw, nw)).astype(np.float32)
    if name == "SNMF":
        V -= 0.4
    if name == "BNMF":
        V = (V < 0.3).astype(np.float32)
    mdl, o = cls(V, num_bases=k, **kw), ocls(V.astype(np.float64), num_bases=k, **kw)
    if name == "RNMF":
        np.random.seed(3); o.factorize(niter=1)
        np.random.seed(3); mdl.factorize(niter=1)
    else:
        W0, H0 = rs.random_sample((mw, k)), rs.random_sample((k, nw))
        mdl.W, mdl.H = W0.copy(), H0.copy(); o.W, o.H = W0.copy(), H0.copy()
        mdl.factorize(niter=1); o.factorize(niter=1)
    eW = np.linalg.norm(mdl.W - o.W) / np.linalg.norm(o.W); eH = np.linalg.norm(mdl.H - o.H) / np.linalg.norm(o.H)
    # NMFALS: the QPs amplify the float32 rounding of their right-hand sides by the conditioning of H H^T / W^T W (all-positive
    # random factors: about 50 each way), RNMF: |x| - x of a sum of 300 000 mixed-sign terms cancels; the fit (ferr) is what holds
    tolW, tolH = {"NMFALS": (3e-4, 2e-2), "RNMF": (2e-3, 2e-5)}.get(name, (2e-5, 2e-5))
    ok = eW < tolW and eH < tolH and abs(mdl.ferr[-1] - o.ferr[-1]) <= 1e-6 * o.ferr[-1]
    print("%-6s n = %7d k = %3d: one iteration vs the oracle class: relW %.2e relH %.2e ferr rel %.1e %s" %
          (name, nw, k, eW, eH, abs(mdl.ferr[-1] - o.ferr[-1]) / o.ferr[-1], "" if ok else "BAD"), flush=True)
    bad += not ok
    mdl._ctx.close()
print("bad %d" % bad)
