#!/usr/bin/env python3
"""Streamed (out-of-core) NMF iteration rate: tools/stream_bench.py [m n k tile_rows iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymf_amd import _lib

a = [int(x) for x in sys.argv[1:]]
m, n, k, rows, iters = (a + [1048576, 256, 64, 65536, 5][len(a):])[:5]
V = np.random.RandomState(1234).random_sample((m, n)).astype(np.float32)
ctx = _lib.Context(_lib.ALGO_NMF, m, n, k)
ctx.fill_w_uniform(42, 0)
ctx.fill_h_uniform(43)
for it in range(iters + 1):
    if it == 1:
        ctx.synchronize(); t0 = time.perf_counter()
    ctx.stream_begin(max_tile_rows=rows)
    for r0 in range(0, m, rows):
        ctx.stream_tile(r0, V[r0:r0 + rows])
    ferr, nd = ctx.stream_end()
ctx.synchronize()
dt = (time.perf_counter() - t0) / iters
print("streamed m=%d n=%d k=%d tile=%d rows: %.2f ms/iter = %.1f GB/s of V over PCIe, ferr=%.4f" %
      (m, n, k, rows, dt * 1e3, m * n * 4 / dt / 1e9, ferr))
