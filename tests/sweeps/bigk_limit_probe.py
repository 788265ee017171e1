"""ad-hoc: every class beyond 1 024 bases against the float64 oracles (is the 1 024 limit of pmf_ctx_create a real one?)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pymf_amd
from pymf_amd.rnmf import RNMF
from pymf_amd.bnmf import BNMF
import oracle


def rel(a, b):
    return np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-300)


def run(cls_name, m, n, k, niter, lo=0.0, **kw):
    rs = np.random.RandomState(m + n + k)
    V = (rs.random_sample((m, n)) - lo).astype(np.float32)
    W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
    o = getattr(oracle, cls_name + "Oracle")(V.astype(np.float64), num_bases=k, **kw)
    cls = {"RNMF": RNMF, "BNMF": BNMF}.get(cls_name) or getattr(pymf_amd, cls_name)
    mdl = cls(V, num_bases=k, **kw)
    if cls_name == "RNMF":
        np.random.seed(5); o.factorize(niter=niter)
        np.random.seed(5); t0 = time.time(); mdl.factorize(niter=niter)
    else:
        o.W, o.H = W0.copy(), H0.copy()
        mdl.W, mdl.H = W0.copy(), H0.copy()
        o.factorize(niter=niter)
        t0 = time.time(); mdl.factorize(niter=niter)
    dt = time.time() - t0
    print(cls_name, (m, n, k), "relW %.2e relH %.2e ferr rel %.1e  %.2fs" %
          (rel(mdl.W, o.W), rel(mdl.H, o.H), np.max(np.abs(mdl.ferr - o.ferr) / o.ferr), dt), flush=True)


if __name__ == "__main__":
    for k in (1500, 2304):
        run("NMF", 3000, 2600, k, 3)
        run("BNMF", 3000, 2600, k, 3)
        run("SNMF", 3000, 2600, k, 2, lo=0.5)
        run("RNMF", 3000, 2600, k, 2, lamb=1.0)
    run("NMFALS", 260, 220, 1100, 2)
