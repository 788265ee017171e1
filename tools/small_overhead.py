import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import pymf_amd
V = np.random.RandomState(20260101).random_sample((100, 50)).astype(np.float32)
np.random.seed(42)
t = time.perf_counter(); m = pymf_amd.NMF(V, num_bases=4); m.factorize(niter=50); print("cfg1 first factorize(50): %.2f ms" % ((time.perf_counter() - t) * 1e3), m.last_call_ms)
for _ in range(3):
    t = time.perf_counter(); m.factorize(niter=50); print("  again: %.2f ms" % ((time.perf_counter() - t) * 1e3), {k: round(v, 3) for k, v in m.last_call_ms.items()})
t = time.perf_counter(); m2 = pymf_amd.NMF(V, num_bases=4); m2.factorize(niter=50); print("second object first factorize(50): %.2f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter()
for _ in range(20): m2.update_w(); m2.update_h()
print("20 x (update_w + update_h) hooks: %.2f ms" % ((time.perf_counter() - t) * 1e3))
