import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from pymf_amd import _lib
shape, k = (5000, 128), 128
rs = np.random.RandomState(shape[1] + k)
Vs = sp.random(shape[0], shape[1], density=0.02, format="csr", dtype=np.float32, random_state=rs)
W0 = rs.random_sample((shape[0], k)).astype(np.float32)
H0 = (rs.random_sample((k, shape[1])) + 0.1).astype(np.float32)
def run(gram, h64, pipe=None, niter=5):
    c = _lib.Context(_lib.ALGO_SNMF, shape[0], shape[1], k)
    c.set_v_csr(Vs.indptr, Vs.indices, Vs.data)
    c.set_w(W0); c.set_h(H0)
    c.set_option("snmf_gram", gram); c.set_option("snmf_h64", h64)
    if pipe is not None: c.set_option("snmf_w_pipe", pipe)
    c.factorize(niter, compute_err=False)
    W, H = c.get_w(), c.get_h()
    Hd = np.zeros((k, shape[1])); c.get_h_into(Hd)
    c.close()
    return W, H, Hd
for h64 in (1, 0):
    a = run(1, h64); b = run(1, h64); c2 = run(2, h64); d = run(2, h64); e = run(2, h64, pipe=0)
    print("h64", h64, "gram1 twice W", np.array_equal(a[0], b[0]), "H", np.array_equal(a[1], b[1]), "Hd", np.array_equal(a[2], b[2]))
    print("   gram2 twice W", np.array_equal(c2[0], d[0]), "H", np.array_equal(c2[1], d[1]))
    print("   gram1 vs gram2: W", np.array_equal(a[0], c2[0]), (a[0] != c2[0]).mean(), "H", np.array_equal(a[1], c2[1]), "Hd", np.array_equal(a[2], c2[2]), np.abs(a[2]-c2[2]).max())
    print("   gram1 vs gram2 nopipe: W", np.array_equal(a[0], e[0]), "H", np.array_equal(a[1], e[1]), "Hd", np.array_equal(a[2], e[2]))
    for n in (1, 2, 3):
        x, y = run(1, h64, niter=n), run(2, h64, niter=n)
        print("   niter", n, "W", np.array_equal(x[0], y[0]), "H", np.array_equal(x[1], y[1]), "Hd", np.array_equal(x[2], y[2]))
