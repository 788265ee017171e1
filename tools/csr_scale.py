#!/usr/bin/env python3
"""SNMF CSR iteration time vs row count (fixed-cost check): tools/csr_scale.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymf_amd import _lib
n, k = 128, 128
for m in (65536, 262144, 524288, 1048576, 2097152):
    rs = np.random.RandomState(1234)
    nnz_row = np.minimum(rs.poisson(0.01 * n, size=m).astype(np.int64), n)
    indptr = np.concatenate([[0], np.cumsum(nnz_row)]); nnz = int(indptr[-1])
    indices = rs.randint(0, n, size=nnz).astype(np.int32); vals = rs.random_sample(nnz).astype(np.float32)
    ctx = _lib.Context(_lib.ALGO_SNMF, m, n, k)
    ctx.set_v_csr(indptr, indices, vals); ctx.fill_w_uniform(42); ctx.fill_h_uniform(43)
    ctx.factorize(2, compute_err=False)
    t = time.time(); ctx.factorize(20, compute_err=False); dt = (time.time() - t) / 20
    print("m=%8d: %.3f ms/iter" % (m, dt * 1e3), flush=True)
    ctx.close()
