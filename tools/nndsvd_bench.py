#!/usr/bin/env python3
"""Time pmf_nndsvd_init (device-filled V) at a few shapes: tools/nndsvd_bench.py [m n k]..."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymf_amd import _lib

shapes = [(1048576, 256, 64), (65536, 512, 32), (262144, 1024, 64), (4096, 64, 16)]
if len(sys.argv) >= 4:
    shapes = [tuple(int(x) for x in sys.argv[1:4])]
for m, n, k in shapes:
    ctx = _lib.Context(_lib.ALGO_NMF, m, n, k)
    ctx.fill_v_uniform(1234, 0)
    ctx.nndsvd_init()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.nndsvd_init()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ferr, done, _ = ctx.factorize(3, True, True, True)
    print("m=%d n=%d k=%d: nndsvd_init %.1f ms; first NMF errors %s" % (m, n, k, dt * 1e3, ferr[:done]))
    ctx.close()
