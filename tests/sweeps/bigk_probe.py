"""ad-hoc: SNMF / NMFALS / RNMF beyond their round-1 base limits against the float64 oracles."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pymf_amd
import oracle

def rel(a, b):
    return np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-300)

def run(cls_name, m, n, k, niter, sparse=False, hooks=False, gram=None, lo=0.0):
    rs = np.random.RandomState(m + n + k)
    V = rs.random_sample((m, n)) - lo
    if sparse:
        V = V * (rs.random_sample((m, n)) < 0.1)
    V = V.astype(np.float32)
    W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
    o = getattr(oracle, cls_name + "Oracle")(V.astype(np.float64), num_bases=k)
    o.W, o.H = W0.copy(), H0.copy()
    data = V
    if sparse:
        import scipy.sparse as sp
        data = sp.csr_matrix(V)
    mdl = getattr(pymf_amd, cls_name)(data, num_bases=k)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    t0 = time.time()
    if hooks:
        for _ in range(niter):
            mdl.update_w(); mdl.update_h()
            o.update_w(); o.update_h()
        fe = "hooks"
    else:
        if gram is not None:
            mdl._context().set_option("snmf_gram", gram)
        ce = not sparse
        mdl.factorize(niter=niter, compute_err=ce)
        o.factorize(niter=niter, compute_err=ce)
        fe = "ferr rel %.1e" % (np.max(np.abs(mdl.ferr - o.ferr) / o.ferr)) if ce else "no err"
    print(cls_name, (m, n, k), "sparse" if sparse else "dense", "gram=%s" % gram, "relW %.2e relH %.2e %s  %.2fs" %
          (rel(mdl.W, o.W), rel(mdl.H, o.H), fe, time.time() - t0), flush=True)

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "snmf"
    if what == "snmf":
        run("SNMF", 3000, 400, 160, 4, lo=0.3)
        run("SNMF", 3000, 400, 160, 4, lo=0.3, gram=0)
        run("SNMF", 3000, 400, 160, 3, lo=0.3, hooks=True)
        run("SNMF", 2000, 700, 300, 3, lo=0.5)
        run("SNMF", 2000, 300, 200, 3, sparse=True)
        run("SNMF", 2000, 300, 200, 3, sparse=True, hooks=True)
        run("SNMF", 1500, 1200, 520, 2, lo=0.5)
    elif what == "als":
        run("NMFALS", 600, 200, 80, 3)
        run("NMFALS", 600, 300, 128, 3)
        run("NMFALS", 500, 400, 200, 2)
        run("NMFALS", 500, 300, 100, 2, hooks=True)
    elif what == "rnmf":
        from pymf_amd.rnmf import RNMF
        for (m, n, k, niter) in [(1000, 300, 160, 4), (700, 520, 260, 3)]:
            rs = np.random.RandomState(m + k)
            V = rs.random_sample((m, n)).astype(np.float32)
            V.flat[rs.randint(0, V.size, size=V.size // 300)] += 5.0
            np.random.seed(5)
            mdl = RNMF(V, num_bases=k, lamb=1.0)
            mdl.factorize(niter=niter)
            np.random.seed(5)
            o = oracle.RNMFOracle(V, num_bases=k, lamb=1.0)
            o.factorize(niter=niter)
            print("RNMF", (m, n, k), "relW %.2e relH %.2e relS %.2e ferr rel %.1e" %
                  (rel(mdl.W, o.W), rel(mdl.H, o.H), rel(mdl.S, o.S), np.max(np.abs(mdl.ferr - o.ferr) / o.ferr)), flush=True)
