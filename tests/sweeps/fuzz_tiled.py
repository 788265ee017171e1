#!/usr/bin/env python3
"""Random shapes through the any-shape two-pass kernels (force_tiled) with the stream kernels on and off: W, H must be
bit-identical between the two, and agree with the float64 oracle; ragged row counts, every base-count class."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pymf_amd import _lib
from oracle import NMFOracle

rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ntrial = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for trial in range(ntrial):
    m = int(rs.choice([1, 17, 64, 65, 200, 255, 256, 257, 1000, 4097, 20000]))
    n = int(rs.choice([64, 100, 128, 129, 192, 256, 300, 384, 512, 640, 1024, 1100]))
    k = int(rs.choice([1, 3, 16, 17, 32, 33, 48, 64, 65, 100, 128, 129, 200, 256, 300]))
    V = rs.random_sample((m, n)).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32); H0 = rs.random_sample((k, n)).astype(np.float32)
    outs = []
    try:
        for stream in (1, 0):
            c = _lib.Context(_lib.ALGO_NMF, m, n, k)
            c.set_option("force_tiled", 1)
            c.set_option("rowgemm_stream", stream); c.set_option("colgemm_stream", stream)
            c.set_v_dense(V); c.set_w(W0); c.set_h(H0)
            for _ in range(3):
                c.update_w(); c.update_h()
            outs.append((c.get_w(), c.get_h())); c.close()
        same = np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        o = NMFOracle(V, num_bases=k); o.W, o.H = W0.astype(np.float64), H0.astype(np.float64)
        for _ in range(3):
            o.update_w(); o.update_h()
        e = max(np.linalg.norm(outs[0][0] - o.W) / np.linalg.norm(o.W), np.linalg.norm(outs[0][1] - o.H) / np.linalg.norm(o.H))
        flag = "" if (same and e < 2e-5) else "  <<<<<"
        if flag: bad += 1
        print(m, n, k, "bit-identical" if same else "DIFFERENT", "vs oracle %.1e" % e, flag, flush=True)
    except Exception as ex:
        bad += 1; print(m, n, k, "EXC", type(ex).__name__, str(ex)[:120], flush=True)
print("bad", bad)
