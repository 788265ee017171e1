#!/usr/bin/env python3
"""Where does the first factorize() of a fresh object spend its wall time?  
   python tools/first_call_probe.py [m n k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pymf_amd
from pymf_amd import _lib
from pymf_amd.nmf import _fingerprint

m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1048576, 256, 64)
V = np.random.RandomState(1).random_sample((m, n)).astype(np.float32)
W = np.random.RandomState(2).random_sample((m, k))
H = np.random.RandomState(3).random_sample((k, n))


def t(label, fn):
    t0 = time.perf_counter()
    r = fn()
    print("%-46s %8.2f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True)
    return r


_lib.load()
for rep in range(2):
    print("---- pass %d" % rep)
    ctx = t("Context()", lambda: _lib.Context(_lib.ALGO_NMF, m, n, k))
    t("digest V (1 GiB float32)", lambda: _fingerprint(V))
    t("set_v_dense (float32)", lambda: ctx.set_v_dense(V))
    t("set_v_dense again", lambda: ctx.set_v_dense(V))
    t("digest W (float64)", lambda: _fingerprint(W))
    t("set_w (float64, rounded on the device)", lambda: ctx.set_w(W))
    t("set_w again", lambda: ctx.set_w(W))
    W32 = t("host W.astype(float32)", lambda: W.astype(np.float32))
    t("set_w (float32)", lambda: ctx.set_w(W32))
    t("set_h (float64)", lambda: ctx.set_h(H))
    t("factorize(50)", lambda: ctx.factorize(50, compute_err=False))
    print("   device loop %.2f ms" % ctx.last_loop_ms())
    Wo = np.empty((m, k))
    t("get_w_into (float64)", lambda: ctx.get_w_into(Wo))
    t("get_w (float32)", lambda: ctx.get_w())
    ctx.close()
    mdl = pymf_amd.NMF(V, num_bases=k)
    mdl.W, mdl.H = W.copy(), H.copy()
    t("NMF.factorize(50) first call", lambda: mdl.factorize(niter=50, compute_err=False))
    print("   ", dict((a, round(b, 2)) for a, b in mdl.last_call_ms.items()))
    t("NMF.factorize(50) second call", lambda: mdl.factorize(niter=50, compute_err=False))
    t("read .W (float64 host array refreshed)", lambda: mdl.W)
    mdl._ctx.close()
