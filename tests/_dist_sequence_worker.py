"""world_size-N worker: random SEQUENCES of calls (the single-process sweep tests/sweeps/fuzz_sequences.py, under a
multi-rank world) on row-sharded objects, against the UNSHARDED oracle twin.  Every rank draws the same sequence from the same
seed; the calls that sum over the ranks (factorize, update_h, frobenius_norm) are collective by construction, the others are
local.  argv: [--seed S] [--cases C].  The ranks may share a GPU (PYMF_DIST_TRANSPORT = host | ipc)."""
import copy
import os
import pickle
import sys
import warnings
import logging

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pymf_amd import dist          # noqa: E402
import pymf_amd                    # noqa: E402
from pymf_amd.bnmf import BNMF     # noqa: E402
import oracle                      # noqa: E402

warnings.simplefilter("ignore")
logging.disable(logging.CRITICAL)


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def gather_rows(block):
    parts = dist.allgather_bytes(np.ascontiguousarray(block, dtype=np.float64).tobytes())
    return np.concatenate([np.frombuffer(p, dtype=np.float64).reshape(-1, block.shape[1]) for p in parts], axis=0)


def rel(a, b):
    return np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-300)


def main():
    w = dist.init_from_env()
    assert w.size > 1
    seed, ncase = arg("--seed", 0), arg("--cases", 12)
    rs = np.random.RandomState(seed)                 # the SAME stream on every rank
    classes = [("NMF", pymf_amd.NMF, oracle.NMFOracle, 2e-5), ("SNMF", pymf_amd.SNMF, oracle.SNMFOracle, 2e-4),
               ("NMFALS", pymf_amd.NMFALS, oracle.NMFALSOracle, 2e-3), ("BNMF", BNMF, oracle.BNMFOracle, 5e-5)]
    bad = 0
    for case in range(ncase):
        name, cls, ocls, tol = classes[int(rs.randint(len(classes)))]
        m = int(rs.choice([37, 260, 1003, 4100])); n = int(rs.choice([64, 100, 256, 300])); k = int(rs.choice([1, 3, 8, 16, 33, 64]))
        if name == "NMFALS":
            k = min(k, 8)
        if name == "SNMF":
            k = max(1, min(k, n // 2))
        V = rs.random_sample((m, n)).astype(np.float32) - (0.4 if name == "SNMF" else 0.0)
        if name == "BNMF":
            V = (V < 0.35).astype(np.float32)
        lo, hi = w.row_range(m)
        use_cls, variant = cls, []
        if rs.randint(4) == 0:                # a subclass that overrides a hook: factorize() runs the hook loop
            class Hooked(cls):
                def update_h(self):
                    cls.update_h(self)
            use_cls = Hooked; variant.append("hooked")
        a, o = use_cls(V[lo:hi].copy(), num_bases=k), ocls(V.astype(np.float64), num_bases=k)
        if rs.randint(4) == 0:
            a.stream_rows = int(rs.choice([64, 256])); variant.append("stream_rows=%d" % a.stream_rows)
        W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
        a.W, a.H = W0[lo:hi].copy(), H0.copy(); o.W, o.H = W0.copy(), H0.copy()
        log, ok = [], True
        if name == "BNMF":
            kw = dict(niter=int(rs.randint(1, 5)))
            a.factorize(**kw); o.factorize(**kw); log.append("factorize(%s)" % kw)
        for step in range(int(rs.randint(4, 10))):
            op = int(rs.randint(11))
            if op <= 2:
                kw = dict(niter=int(rs.randint(1, 6)), compute_w=bool(rs.randint(2)), compute_h=bool(rs.randint(2)), compute_err=bool(rs.randint(2)))
                log.append("factorize(%s)" % kw)
                a.factorize(**kw); o.factorize(**kw)
                if kw["compute_err"]:
                    L = min(len(a.ferr), len(o.ferr))
                    floor = 5e-6 * max(np.linalg.norm(V), np.linalg.norm(o.W) * np.linalg.norm(o.H))
                    same = np.allclose(a.ferr[:L], o.ferr[:L], rtol=1e-4, atol=floor)
                    if len(a.ferr) != len(o.ferr):         # stationary to float32 noise (see the single-process sweep)
                        tail = np.concatenate([np.abs(np.diff(a.ferr[L - 1:])), np.abs(np.diff(o.ferr[L - 1:])), np.abs(a.ferr[L - 1:L] - o.ferr[L - 1:L])])
                        same = same and tail.max() <= floor
                        if same:
                            o.W, o.H = gather_rows(a.W), np.array(a.H, dtype=np.float64)
                            if name == "BNMF":
                                o._lamb_W, o._lamb_H = a._lamb_W, a._lamb_H
                    lens = dist.allgather_bytes(np.array([len(a.ferr)], dtype=np.int64).tobytes())
                    if len(set(lens)) != 1:
                        same = False; log.append("ranks left the loop at different iterations")
                    if not same:
                        ok = False; log.append("ferr %s vs %s" % (a.ferr, o.ferr))
            elif op == 3:
                log.append("update_w"); a.update_w(); o.update_w()
            elif op == 4:
                log.append("update_h"); a.update_h(); o.update_h()
            elif op == 5:
                log.append("frobenius_norm")
                fa, fo = a.frobenius_norm(), o.frobenius_norm()
                if abs(fa - fo) > 1e-4 * fo + 5e-6 * max(np.linalg.norm(V), np.linalg.norm(o.W) * np.linalg.norm(o.H)):
                    ok = False; log.append("frobenius %r vs %r" % (fa, fo))
            elif op == 6:
                log.append("assign W"); Wn = o.W * (1.0 + 0.1 * rs.random_sample(o.W.shape)); a.W = Wn[lo:hi].copy(); o.W = Wn.copy()
            elif op == 7:
                log.append("edit H in place"); i, j = int(rs.randint(k)), int(rs.randint(n))
                hv = a.H; hv[i, j] = hv[i, j] * 1.5 + 0.01; o.H[i, j] = o.H[i, j] * 1.5 + 0.01
            elif op == 8:
                log.append("edit data in place"); i, j = int(rs.randint(m)), int(rs.randint(n))
                if lo <= i < hi:
                    a.data[i - lo, j] += 0.25
                o.data[i, j] += 0.25
            elif op == 9:
                log.append("replace data"); Vn = (o.data * (1.0 + 0.05 * rs.random_sample(o.data.shape))).astype(np.float32)
                a.data = Vn[lo:hi].copy(); o.data = Vn.astype(np.float64)
            else:
                how = int(rs.randint(3))
                if "hooked" in variant and how == 2:
                    how = 1                   # (a class defined inside a function does not pickle)
                log.append(["copy.copy", "copy.deepcopy", "pickle"][how])
                a = copy.copy(a) if how == 0 else copy.deepcopy(a) if how == 1 else pickle.loads(pickle.dumps(a))
            eW, eH = rel(gather_rows(a.W), o.W), rel(a.H, o.H)
            hs = dist.allgather_bytes(np.ascontiguousarray(a.H).tobytes())
            if not all(h == hs[0] for h in hs):
                ok = False; log.append("H differs between the ranks")
            if not (eW < tol and eH < tol):
                ok = False; log.append("relW %.2e relH %.2e" % (eW, eH))
            oks = dist.allgather_bytes(np.array([int(ok)], dtype=np.int64).tobytes())
            ok = all(np.frombuffer(b, dtype=np.int64)[0] for b in oks)      # (every rank stops the case together)
            if not ok:
                break
        if not ok:
            bad += 1
            print("rank %d BAD case %d: %s %s %dx%d k=%d: %s" % (w.rank, case, name, variant, m, n, k, " -> ".join(log[-8:])), flush=True)
        try:
            a._ctx.close()
        except Exception:
            pass
    print("rank %d: seed %d, %d cases, bad %d" % (w.rank, seed, ncase, bad), flush=True)
    dist.barrier()
    dist.shutdown()
    assert bad == 0
    print("rank %d ok" % w.rank)


if __name__ == "__main__":
    main()
