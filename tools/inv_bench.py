"""time of one Gram-space SNMF iteration at k = n = 128 (k_inverse_spd_mfma is its longest kernel)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymf_amd import _lib
import bench
m, n, k = 65536, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 128
ip, ix, vv = bench.gen_csr(m, n, 0.01, 0, m, True)
c = _lib.Context(_lib.ALGO_SNMF, m, n, k)
c.set_v_csr(ip, ix, vv); c.fill_w_uniform(42); c.fill_h_uniform(43)
c.factorize(50, compute_err=False)
t = time.time(); c.factorize(500, compute_err=False); dt = time.time() - t
print("k=%d: %.1f us per Gram-space iteration (incl. 1/500 of the W pass)" % (k, dt / 500 * 1e6))
