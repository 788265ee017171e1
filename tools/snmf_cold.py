import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pymf_amd import _lib
for (m, n, k) in ((1048576, 256, 64), (262144, 1024, 64)):
    ctx = _lib.Context(_lib.ALGO_SNMF, m, n, k)
    ctx.fill_v_uniform(1234); ctx.fill_w_uniform(42); ctx.fill_h_uniform(43)
    ctx.synchronize()
    t = time.time(); ctx.factorize(20, compute_err=False); ctx.synchronize(); dt = time.time() - t
    print("SNMF dense %dx%d k=%d: cold factorize(20) %.2f ms (device loop %.2f ms)" % (m, n, k, dt * 1e3, ctx.last_loop_ms()))
    t = time.time(); ctx.factorize(20, compute_err=False); ctx.synchronize(); dt = time.time() - t
    print("   second call %.2f ms" % (dt * 1e3))
