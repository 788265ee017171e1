#!/usr/bin/env python3
"""Long loops and many contexts: device memory and host RSS must come back (leak check), results must stay finite."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import ctypes
import pymf_amd
from pymf_amd import _lib

hip = ctypes.CDLL("libamdhip64.so")


def free_mib():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**20


def rss_mib():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0


rs = np.random.RandomState(0)
V = rs.random_sample((4096, 256)).astype(np.float32)
bad = 0
# 1. contexts created and closed over and over, every class
_lib.load()
m0 = None
for rep in range(60):
    for cls, kw in ((pymf_amd.NMF, {}), (pymf_amd.SNMF, {}), (pymf_amd.NMFALS, {})):
        mdl = cls(V if cls is not pymf_amd.SNMF else V - 0.5, num_bases=16)
        mdl.factorize(niter=3)
        if not np.all(np.isfinite(mdl.W)) or not np.all(np.isfinite(mdl.H)):
            bad += 1
        mdl._ctx.close(); mdl._ctx = None
    if rep == 4:
        m0, r0 = free_mib(), rss_mib()
m1, r1 = free_mib(), rss_mib()
print("contexts: free device memory %.0f -> %.0f MiB, max RSS %.0f -> %.0f MiB" % (m0, m1, r0, r1))
if m0 - m1 > 64 or r1 - r0 > 256:
    bad += 1
# 2. one object, long free-running loops
mdl = pymf_amd.NMF(V, num_bases=16)
f0 = None
for rep in range(4):
    t0 = time.time()
    mdl.factorize(niter=20000, compute_err=(rep % 2 == 0))
    if rep == 0:
        f0, r0 = free_mib(), rss_mib()
    print("factorize(20000) %.2f s, len(ferr) %d, ferr[-1] %.6g" % (time.time() - t0, len(mdl.ferr) if hasattr(mdl, "ferr") and mdl.ferr is not None else 0,
                                                              mdl.ferr[-1] if rep % 2 == 0 else float("nan")), flush=True)
    if not np.all(np.isfinite(mdl.W)):
        bad += 1
f1, r1 = free_mib(), rss_mib()
print("long loops: free device memory %.0f -> %.0f MiB, max RSS %.0f -> %.0f MiB" % (f0, f1, r0, r1))
if f0 - f1 > 64 or r1 - r0 > 256:
    bad += 1
print("bad %d" % bad)
