#!/usr/bin/env python3
"""current RSS over many context create / close cycles (is anything leaked per context?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pymf_amd import _lib


def rss():
    return int(open("/proc/self/statm").read().split()[1]) * 4096 / 2**20


V = np.random.RandomState(0).random_sample((4096, 256)).astype(np.float32)
W = np.random.RandomState(1).random_sample((4096, 16)).astype(np.float32)
H = np.random.RandomState(2).random_sample((16, 256)).astype(np.float32)
_lib.load()
for algo, name in ((_lib.ALGO_NMF, "NMF"), (_lib.ALGO_SNMF, "SNMF"), (_lib.ALGO_NMFALS, "NMFALS")):
    pts = []
    for rep in range(301):
        ctx = _lib.Context(algo, 4096, 256, 16)
        ctx.set_v_dense(V); ctx.set_w(W); ctx.set_h(H)
        ctx.factorize(2)
        ctx.close()
        if rep % 100 == 0:
            pts.append(rss())
    print(name, "RSS MiB after 1 / 101 / 201 / 301 contexts:", " ".join("%.0f" % p for p in pts), flush=True)
pts = []
for rep in range(301):
    ctx = _lib.Context(_lib.ALGO_NMF, 4096, 256, 16)
    ctx.close()
    if rep % 100 == 0:
        pts.append(rss())
print("create + close only:", " ".join("%.0f" % p for p in pts))
