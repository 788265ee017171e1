#!/usr/bin/env python3
"""update_w() + update_h() called singly (the reference's plugin API) at cfg4 size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymf_amd import _lib
m, n, k = 1048576, 256, 64
for algo, name in ((_lib.ALGO_NMF, "NMF"), (_lib.ALGO_SNMF, "SNMF")):
    ctx = _lib.Context(algo, m, n, k)
    ctx.fill_v_uniform(1234); ctx.fill_w_uniform(42); ctx.fill_h_uniform(43)
    for _ in range(3):
        ctx.update_w(); ctx.update_h()
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(20):
        ctx.update_w(); ctx.update_h()
    ctx.synchronize(); dt = (time.perf_counter() - t) / 20
    print("%s hooks: update_w + update_h = %.3f ms per pair" % (name, dt * 1e3))
    ctx.close()
