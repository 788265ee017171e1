"""What does a factorize() call of K iterations cost per iteration -- as a function of K (clock ramp behind the idle gap between two calls),
with and without the live HIP events around the dominant kernel (pmf_profile_enable)?   usage: loop_probe.py m n k"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymf_amd import _lib
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = _lib.Context(0, m, n, k)
ctx.fill_v_uniform(1234); ctx.fill_w_uniform(42); ctx.fill_h_uniform(43)
ctx.factorize(3, compute_err=False)
ctx.factorize(int(0.3 / 65e-6), compute_err=False)          # 0.3 s pre-roll
for prof in (False, True, False, True):
    ctx.profile_enable(prof)
    for K in (20, 200, 2000, 20, 200, 2000):
        ctx.synchronize()
        t = time.perf_counter(); ctx.factorize(K, compute_err=False); ctx.synchronize(); dt = time.perf_counter() - t
        line = "events %-5s K=%-5d wall %.2f us/iter, device loop %.2f us/iter" % (prof, K, dt / K * 1e6, ctx.last_loop_ms() / K * 1e3)
        if prof:
            ms = ctx.kernel_launch_ms()
            line += "; kernel by events: mean %.2f first %.2f last %.2f us (n=%d)" % (ms.mean() * 1e3, ms[0] * 1e3, ms[-1] * 1e3, len(ms))
            ctx.profile_enable(True)
        print(line)
