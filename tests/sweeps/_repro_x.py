import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, warnings, logging
warnings.simplefilter("ignore"); logging.disable(logging.CRITICAL)
import pymf_amd, oracle
def rel(a,b): return np.linalg.norm(a-b)/np.linalg.norm(b)
for cls, ocls in ((pymf_amd.NMF, oracle.NMFOracle), (pymf_amd.NMFALS, oracle.NMFALSOracle), (pymf_amd.SNMF, oracle.SNMFOracle)):
  for layout in ("C float32", "F float32", "C float64", "F float64", "strided float32"):
    for sr in (0, 64):
        m, n, k = 3000, 300, 4
        rs = np.random.RandomState(1)
        Vn = rs.random_sample((m, n)).astype(np.float32) - (0.4 if cls is pymf_amd.SNMF else 0.0)
        Vd = {"C float32": Vn.copy(), "F float32": np.asfortranarray(Vn), "C float64": Vn.astype(np.float64), "F float64": np.asfortranarray(Vn.astype(np.float64)),
              "strided float32": np.repeat(Vn, 2, axis=1)[:, ::2]}[layout]
        W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
        a, o = cls(Vd, num_bases=k), ocls(Vn.astype(np.float64), num_bases=k)
        if sr: a.stream_rows = sr
        a.W, a.H = W0.copy(), H0.copy(); o.W, o.H = W0.copy(), H0.copy()
        a.factorize(niter=3); o.factorize(niter=3)
        e = rel(a.W, o.W)
        print(cls.__name__, layout, "stream_rows=%d" % sr, "relW %.2e relH %.2e ferr rel %.1e %s" % (e, rel(a.H, o.H), abs(a.ferr[-1]-o.ferr[-1])/o.ferr[-1], "BAD" if e > 1e-3 else ""), flush=True)
