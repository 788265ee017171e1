#!/usr/bin/env python3
"""Diagnostic: passes, system sizes and section ticks of k_nnqp_quad at cfg3, per iteration.  Needs a counting build:
  (cd pymf_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 \
      -DPMF_QUAD_COUNT pmf_api.hip -o ../../build_ab/lib_quadcount.so -lrccl)
  PMF_LIB=build_ab/lib_quadcount.so python tools/quad_counts.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymf_amd import _lib
m, n, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (262144, 1024, 64)   # k > 64: k_nnqp_wave's counters (sizes / 8)
ctx = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
ctx.fill_v_uniform(1234); ctx.fill_w_uniform(42); ctx.fill_h_uniform(43)
lib = ctx._lib
buf = (ctypes.c_ulonglong * 24)()
lib.pmf_debug_quad_counts(buf, 1)
for it in range(14):
    ctx.factorize(1, compute_err=False); ctx.synchronize()
    lib.pmf_debug_quad_counts(buf, 1)
    v = list(buf)
    tasks, passes = max(v[0], 1), max(v[1], 1)
    print("iter %2d: wave tasks %6d, passes per task %.2f, mean largest system %.1f, mean longest product %.1f, sizes/4 histogram %s"
          % (it, v[0], passes / tasks, v[2] / passes, v[3] / passes, v[4:13]))
    tt = v[16:23]
    print("         ticks per pass: lists %.0f, y product %.0f, gather %.0f, LDL^T %.0f, solves %.0f, z product %.0f, decision %.0f  (sum %.0f)"
          % tuple([x / passes for x in tt] + [sum(tt) / passes]))
