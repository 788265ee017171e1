"""cfg5 shape class (k = n = 128): error of the device SNMF vs the float64 oracle, iteration by
iteration, for dense and CSR input of the same matrix (diagnostic; oracle = test infrastructure)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import pymf_amd
from oracle import SNMFOracle

def rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b)

Vc = sp.random(2048, 128, density=0.01, format="csr", dtype=np.float32, random_state=np.random.RandomState(1234))
Vd = np.asarray(Vc.toarray(), dtype=np.float32)
np.random.seed(42)
W0, H0 = np.random.random((2048, 128)), np.random.random((128, 128))
for it in (1, 2, 3):
    o = SNMFOracle(Vd, num_bases=128); o.W, o.H = W0.copy(), H0.copy(); o.factorize(niter=it, compute_err=False)
    for name, V in (("dense", Vd), ("csr", Vc)):
        m = pymf_amd.SNMF(V, num_bases=128); m.W, m.H = W0.copy(), H0.copy(); m.factorize(niter=it, compute_err=False)
        print("it %d %-5s relW %.3g relH %.3g path %s" % (it, name, rel(m.W, o.W), rel(m.H, o.H), m._ctx.path_name))
    # hooks on the CSR object: update_w alone from the oracle's state
    m = pymf_amd.SNMF(Vc, num_bases=128); m.W, m.H = W0.copy(), o.H.copy(); m.update_w()
    o2 = SNMFOracle(Vd, num_bases=128); o2.W, o2.H = W0.copy(), o.H.copy(); o2.update_w()
    print("      csr update_w hook from the oracle's H: relW %.3g" % rel(m.W, o2.W))

print("---- one iteration, by path ----")
o = SNMFOracle(Vd, num_bases=128); o.W, o.H = W0.copy(), H0.copy(); o.factorize(niter=1, compute_err=False)
for name, V in (("dense", Vd), ("csr", Vc)):
    m = pymf_amd.SNMF(V, num_bases=128); m.W, m.H = W0.copy(), H0.copy(); m.update_w(); m.update_h()
    print("%-5s hooks (update_w, update_h): relW %.3g relH %.3g" % (name, rel(m.W, o.W), rel(m.H, o.H)))
    # H step alone from the ORACLE's W (float32-rounded by the upload)
    m = pymf_amd.SNMF(V, num_bases=128); m.W, m.H = o.W.copy(), H0.copy(); m.update_h()
    print("%-5s update_h from the oracle's W: relH %.3g" % (name, rel(m.H, o.H)))
Wf = o.W.astype(np.float32).astype(np.float64)
P = Wf.T.dot(Vd.astype(np.float64)); S = Wf.T.dot(Wf)
print("scale: |W| %.3g  |P| %.3g  |S| %.3g  |H^T S| %.3g" % (np.abs(o.W).max(), np.abs(P).max(), np.abs(S).max(), np.abs(H0.T.dot(S)).max()))
