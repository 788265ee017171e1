#!/usr/bin/env python3
"""Wall time of pymf_amd.NMF.factorize() through the Python class (host conversions, PCIe) vs the device loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymf_amd
m, n, k = 1048576, 256, 64
V = np.random.RandomState(1234).random_sample((m, n)).astype(np.float32)
np.random.seed(42)
mdl = pymf_amd.NMF(V, num_bases=k)
t = time.time(); mdl.factorize(niter=1); print("first call (uploads V, creates W/H): %.3f s" % (time.time() - t))
for niter in (1, 20, 20):
    t = time.time(); mdl.factorize(niter=niter); dt = time.time() - t
    print("factorize(niter=%d): %.3f s wall, device loop %.1f ms" % (niter, dt, mdl._ctx.last_loop_ms()))
t = time.time(); f = mdl.frobenius_norm(); print("frobenius_norm(): %.3f s" % (time.time() - t))
