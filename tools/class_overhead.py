#!/usr/bin/env python3
"""Wall time of pymf_amd.NMF.factorize() through the Python class (digests, host conversions, PCIe) vs the
device loop, and the speed of the change detector (pmf_host_checksum) by array size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymf_amd
from pymf_amd import _lib

print("host cpu_count", os.cpu_count())
for shape in ((65536, 512), (262144, 256), (1048576, 256)):
    a = np.random.RandomState(1).random_sample(shape).astype(np.float32)
    ts = []
    for _ in range(6):
        t = time.perf_counter(); _lib.host_checksum(a); ts.append(time.perf_counter() - t)
    print("host_checksum %s (%d MiB): " % (shape, a.nbytes >> 20) + " ".join("%.2f" % (x * 1e3) for x in ts) + " ms -> best %.1f GB/s" % (a.nbytes / min(ts) / 1e9))
    del a

m, n, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1048576, 256, 64)))
V = np.random.RandomState(1234).random_sample((m, n)).astype(np.float32)
np.random.seed(42)
mdl = pymf_amd.NMF(V, num_bases=k)
t = time.perf_counter(); mdl.factorize(niter=1); print("first call (uploads V, creates W/H): %.3f s" % (time.perf_counter() - t))
for label, kw in (("default", {}), ("check_data off", {"check_data": False}), ("eager_factors", {"eager_factors": True, "check_data": True})):
    for key in ("check_data", "eager_factors"):
        setattr(mdl, key, kw.get(key, type(mdl).__dict__.get(key, getattr(pymf_amd.NMF, key))))
    for niter in (1, 100):
        t = time.perf_counter(); mdl.factorize(niter=niter, compute_err=False); dt = time.perf_counter() - t
        print("%-16s factorize(niter=%3d): %8.2f ms wall, device loop %8.2f ms" % (label, niter, dt * 1e3, mdl._ctx.last_loop_ms()))
t = time.perf_counter(); w = mdl.W; print("first read of .W: %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); f = mdl.frobenius_norm(); print("frobenius_norm(): %.3f s" % (time.perf_counter() - t))
