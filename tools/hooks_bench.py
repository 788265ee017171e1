"""Per-hook timing of the general (tiled) kernels at cfg4 size: update_w / update_h / direct residual."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymf_amd import _lib
m, n, k = 1048576, 256, 64
ctx = _lib.Context(_lib.ALGO_NMF, m, n, k)
ctx.fill_v_uniform(1234); ctx.fill_w_uniform(42); ctx.fill_h_uniform(43)
for _ in range(3):
    ctx.update_w(); ctx.update_h()
for name, fn in (("update_w (k_rowgemm<4,NMF_W>)", ctx.update_w), ("update_h (k_colgemm<4> + reduce + h_gram)", ctx.update_h)):
    t = time.time()
    for _ in range(20): fn()
    print("%-45s %.3f ms" % (name, (time.time() - t) / 20 * 1e3))
ctx.set_w(ctx.get_w())     # "new" W: stale (P | S), forces the direct residual pass
t = time.time()
for _ in range(10):
    ctx.set_h(ctx.get_h()); ctx.frobenius()
print("%-45s %.3f ms (incl. a 64 KiB H round trip)" % ("frobenius direct (k_resid<4>)", (time.time() - t) / 10 * 1e3))
