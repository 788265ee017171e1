"""ad-hoc: NNDSVD beyond 1024 columns -- error against the float64 closed form and wall time."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import pymf_amd
from oracle import nndsvd_closed_form
from pymf_amd import _lib

def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)

for (m, n, k, kind) in [(3000, 1500, 12, "rand"), (3000, 1500, 12, "lowrank"), (1100, 2500, 8, "lowrank"),
                        (5000, 2048, 32, "lowrank"), (6000, 4096, 16, "lowrank")]:
    rs = np.random.RandomState(m + n + k)
    if kind == "rand":
        V = rs.random_sample((m, n)).astype(np.float32)
    else:
        r = k + 6
        A = rs.random_sample((m, r)) * (1.0 + np.arange(r))[None, ::-1]
        V = (A @ rs.random_sample((r, n)) + 0.05 * rs.random_sample((m, n))).astype(np.float32)
    mdl = pymf_amd.NNDSVD(V, num_bases=k)
    t0 = time.time(); mdl.factorize(); t1 = time.time()
    t2 = time.time(); W, H = nndsvd_closed_form(V, k); t3 = time.time()
    print(m, n, k, kind, "gpu %.2fs cpu-oracle %.2fs relW %.2e relH %.2e" % (t1 - t0, t3 - t2, rel(mdl.W, W), rel(mdl.H, H)), flush=True)
