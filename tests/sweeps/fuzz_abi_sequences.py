#!/usr/bin/env python3
"""Random sequences of C-ABI calls on ONE context (through pymf_amd._lib.Context, no host class in between) against the oracle's
functions: pmf_set_* in any order and dtype, the hooks, pmf_factorize with every flag combination, pmf_frobenius, snapshot /
restore of W, resident passes mixed with STREAMED passes over the same matrix (random tile heights, ragged last tile), options
flipped in between.   python3 tests/sweeps/fuzz_abi_sequences.py [seed] [cases]"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from pymf_amd import _lib
import oracle

warnings.simplefilter("ignore")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rs = np.random.RandomState(seed)
bad = 0


def rel(a, b):
    return np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-300)


for case in range(ncase):
    kind = ["NMF", "NMF", "SNMF", "NMFALS", "BNMF"][int(rs.randint(5))]
    snmf = kind == "SNMF"
    m = int(rs.choice([70, 300, 1000, 2100, 5000])); n = int(rs.choice([64, 100, 256, 300, 520, 1100])); k = int(rs.choice([1, 4, 16, 33, 64, 100, 130]))
    if rs.randint(10) == 0:                   # more columns than one accumulation chain spans (PMF_WIDE_K): the chunked products
        m, n = int(rs.choice([70, 300])), 70000
        k = min(k, 33)
    if snmf:
        k = max(1, min(k, n // 2, m // 2))
    if kind == "NMFALS":
        k = min(k, 8)
    V = rs.random_sample((m, n)).astype(np.float32) - (0.4 if snmf else 0.0)
    if kind == "BNMF":
        V = (V < 0.35).astype(np.float32)
    ocls = {"NMF": oracle.NMFOracle, "SNMF": oracle.SNMFOracle, "NMFALS": oracle.NMFALSOracle, "BNMF": oracle.BNMFOracle}[kind]
    o = ocls(V.astype(np.float64), num_bases=k)
    o.W, o.H = rs.random_sample((m, k)), rs.random_sample((k, n))
    c = _lib.Context({"NMF": _lib.ALGO_NMF, "SNMF": _lib.ALGO_SNMF, "NMFALS": _lib.ALGO_NMFALS, "BNMF": _lib.ALGO_BNMF}[kind], m, n, k)
    c.set_v_dense(V); c.set_w(o.W.copy()); c.set_h(o.H.copy())
    if kind == "BNMF":                        # the penalty weights are plain state at this level (bnmf.py:84-85: x 1.1 per H step)
        o._lamb_W = o._lamb_H = 0.3
        c.set_lambda(0.3, 0.3)
    log, ok = [], True
    snap = None
    tol = {"NMF": 2e-5, "SNMF": 2e-4, "NMFALS": 2e-3, "BNMF": 5e-5}[kind]
    if kind == "NMFALS" and n > 65536:
        tol = 2e-2        # (the QPs amplify the float32 rounding of 70 000-term right-hand sides by the conditioning of W^T W: wide_scan.py)
    for step in range(int(rs.randint(5, 14))):
        op = int(rs.randint(12))
        if op == 11 and not (kind == "NMF" and min(m, n) >= k and n <= 1100):
            op = 2
        if op == 11:
            # the NNDSVD initialiser writes W and H on the device (nndsvd.py:79-108): every cached sum belongs to the old factors.
            # Its own accuracy is another test's subject -- the twin takes the device's factors and the sequence goes on
            log.append("nndsvd_init")
            c.nndsvd_init()
            o.W, o.H = c.get_w().astype(np.float64), c.get_h().astype(np.float64)
        elif op <= 1:
            kw = dict(compute_w=bool(rs.randint(2)), compute_h=bool(rs.randint(2)), compute_err=bool(rs.randint(2)))
            niter = int(rs.randint(1, 6))
            log.append("factorize(%d, %s)" % (niter, kw))
            fe, done, conv = c.factorize(niter, **kw)
            oracle.NMFOracle.factorize(o, niter=niter, **kw)
            if kw["compute_err"]:
                L = min(done, len(o.ferr))
                floor = 5e-6 * max(np.linalg.norm(V), np.linalg.norm(o.W) * np.linalg.norm(o.H))
                if not np.allclose(np.asarray(fe)[:L], o.ferr[:L], rtol=1e-4, atol=floor):
                    ok = False; log.append("ferr %s vs %s" % (np.asarray(fe)[:done], o.ferr))
                if done != len(o.ferr):        # stationary to float32 noise: carry on from the library's state
                    o.W, o.H = c.get_w().astype(np.float64), c.get_h().astype(np.float64)
                    if kind == "BNMF":
                        o._lamb_W, o._lamb_H = c.get_lambda()
        elif op == 2:
            log.append("update_w"); c.update_w(); o.update_w()
        elif op == 3:
            log.append("update_h"); c.update_h(); o.update_h()
        elif op == 4:
            log.append("frobenius")
            fa, fo = c.frobenius(), o.frobenius_norm()
            if abs(fa - fo) > 1e-4 * fo + 5e-6 * max(np.linalg.norm(V), np.linalg.norm(o.W) * np.linalg.norm(o.H)):
                ok = False; log.append("frobenius %r vs %r" % (fa, fo))
        elif op == 5:
            log.append("set_w"); Wn = o.W * (1.0 + 0.1 * rs.random_sample(o.W.shape))
            o.W = Wn.copy(); c.set_w(Wn.astype(np.float32) if rs.randint(2) else Wn)
            if rs.randint(2):
                o.W = Wn.astype(np.float32).astype(np.float64) if False else o.W
        elif op == 6:
            log.append("set_h"); Hn = o.H * (1.0 + 0.1 * rs.random_sample(o.H.shape))
            o.H = Hn.copy(); c.set_h(Hn)
        elif op == 7:
            log.append("set_v"); Vn = (o.data * (1.0 + 0.05 * rs.random_sample(o.data.shape))).astype(np.float32)
            V = Vn; o.data = Vn.astype(np.float64)
            c.set_v_dense(Vn if rs.randint(2) else Vn.astype(np.float64))
        elif op == 8:
            if snap is None or rs.randint(2):
                log.append("snapshot_w"); c.snapshot_w(); snap = o.W.copy()
            else:
                log.append("restore_w"); c.restore_w(); o.W = snap.copy()
        elif op == 9:
            # one STREAMED pass over the same matrix on the same context = one iteration of the loop (nmf.py:183-202)
            rows = int(rs.choice([64, 128, 320, 1024]))
            kw = dict(compute_w=bool(rs.randint(2)), compute_h=bool(rs.randint(2)), compute_err=True)
            log.append("streamed pass(rows=%d, %s)" % (rows, kw))
            c.stream_begin(max_tile_rows=rows, **kw)
            for r0 in range(0, m, rows):
                c.stream_tile(r0, V[r0:r0 + rows])
            fe, nd = c.stream_end()
            oracle.NMFOracle.factorize(o, niter=1, **kw)
            if not nd and abs(fe - o.ferr[-1]) > 1e-4 * o.ferr[-1] + 5e-6 * max(np.linalg.norm(V), np.linalg.norm(o.W) * np.linalg.norm(o.H)):
                ok = False; log.append("streamed ferr %r vs %r" % (fe, o.ferr[-1]))
            c.set_v_dense(V)                  # (the resident calls that follow need V on the device again)
        else:
            name_o, val = [("force_tiled", int(rs.randint(2))), ("snmf_gram", int(rs.randint(3))), ("rowgemm_stream", int(rs.randint(2))),
                           ("colgemm_stream", int(rs.randint(2)))][int(rs.randint(4))]
            log.append("set_option(%s=%d)" % (name_o, val)); c.set_option(name_o, val)
        eW, eH = rel(c.get_w(), o.W), rel(c.get_h(), o.H)
        if not (eW < tol and eH < tol):
            ok = False; log.append("relW %.2e relH %.2e" % (eW, eH))
        if not ok:
            break
    if not ok:
        bad += 1
        print("BAD case %d: %s %dx%d k=%d (%s): %s" % (case, kind, m, n, k, c.path_name, " -> ".join(log[-8:])), flush=True)
    c.close()
print("seed %d: %d cases" % (seed, ncase))
print("bad %d" % bad)
