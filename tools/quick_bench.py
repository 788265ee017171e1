import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymf_amd import _lib
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
niter = int(sys.argv[4]) if len(sys.argv) > 4 else 20
algo = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx = _lib.Context(algo, m, n, k)
if os.environ.get("QB_FORCE_TILED"):
    ctx.set_option("force_tiled", 1)          # the any-shape two-pass kernels on a one-pass kernel's shape
ctx.fill_v_uniform(1234); ctx.fill_w_uniform(42); ctx.fill_h_uniform(43)
if algo == _lib.ALGO_RNMF:
    ctx.set_lambda(0.7, 0.0)
    ctx.rnmf_update_s()                        # rnmf.py:94-98: S exists before the first W step
print("path", ctx.path_name)
ctx.factorize(3, compute_err=False)
for ce in (False, True):
    t = time.time(); ctx.factorize(niter, compute_err=ce); dt = time.time() - t
    print("m=%d n=%d k=%d compute_err=%s: %.3f ms/iter (device loop %.3f ms/iter) -> %.1f it/s" % (m, n, k, ce, dt / niter * 1e3, ctx.last_loop_ms() / niter, niter / dt))
