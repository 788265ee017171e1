#!/usr/bin/env python3
"""Several host threads, one object each, on one GPU at the same time: results must equal the single-threaded ones bit for bit."""
import os, sys, threading, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import pymf_amd


def job(seed, cls_name, out):
    rs = np.random.RandomState(seed)
    m, n, k = int(rs.choice([300, 2048, 5000])), int(rs.choice([64, 200, 256])), int(rs.choice([4, 16, 33]))
    V = rs.random_sample((m, n)).astype(np.float32)
    if cls_name == "SNMF":
        V -= 0.4
    mdl = getattr(pymf_amd, cls_name)(V, num_bases=k)
    mdl.W, mdl.H = rs.random_sample((m, k)), rs.random_sample((k, n))
    mdl.factorize(niter=int(rs.choice([3, 10])))
    out[(seed, cls_name)] = hashlib.sha256(np.ascontiguousarray(mdl.W).tobytes() + np.ascontiguousarray(mdl.H).tobytes() +
                                           np.ascontiguousarray(mdl.ferr).tobytes()).hexdigest()


jobs = [(s, c) for s in range(8) for c in ("NMF", "SNMF", "NMFALS")]
ref = {}
for s, c in jobs:
    job(s, c, ref)
bad = 0
for rnd in range(3):
    got = {}
    ths = [threading.Thread(target=job, args=(s, c, got)) for s, c in jobs]
    for t in ths: t.start()
    for t in ths: t.join()
    diff = [k for k in ref if got.get(k) != ref[k]]
    print("round %d: %d jobs on %d threads, %d differ %s" % (rnd, len(jobs), len(ths), len(diff), diff[:3]), flush=True)
    bad += len(diff)
print("bad %d" % bad)
