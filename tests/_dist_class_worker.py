"""world_size-N worker for the CLASS-level multi-rank path: pymf_amd.NMF / SNMF / RNMF / NNDSVD are
given this rank's block of rows, initialise lazily (rank 0's RNG stream decides), factorize, and the
gathered result must equal the UNSHARDED oracle.

  --fake : no GPU -- pymf_amd._lib.Context is replaced by a test double that does the per-rank math
           with the oracle's functions and sums (W^T V | W^T W) through pymf_amd.dist (CPU suite:
           checks the host logic -- row spans, W0 slices, H0, shapes -- under WORLD_SIZE > 1);
  else   : the real library on the GPU(s); with PYMF_DIST_TRANSPORT=host the ranks may share a GPU.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pymf_amd import dist, _lib          # noqa: E402
import pymf_amd                          # noqa: E402
import oracle                            # noqa: E402

FAKE = "--fake" in sys.argv
DIGEST = "--digest" in sys.argv        # print digests of the factors: the one-shot IPC all-reduce vs the host transport, bit for bit


def digest(name, a):
    if DIGEST:
        import hashlib
        print("digest %s %s" % (name, hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()))


class FakeContext(object):
    """Per-rank math of one context with NumPy (float64) + the cross-rank sum through dist."""

    def __init__(self, algo, m, n, k, device=0, rank=0, nranks=1, nccl_id=None):
        assert algo == _lib.ALGO_NMF
        self.m, self.n, self.k = m, n, k
        self.path_name = "fake"

    def set_v_dense(self, V):
        assert V.shape == (self.m, self.n)
        self.V = np.asarray(V, dtype=np.float64)

    def set_w(self, W):
        assert W.shape == (self.m, self.k), (W.shape, self.m, self.k)
        self.W = np.array(W, dtype=np.float64)

    def set_h(self, H):
        assert H.shape == (self.k, self.n)
        self.H = np.array(H, dtype=np.float64)

    def get_w(self):
        return self.W.astype(np.float32)

    def get_h(self):
        return self.H.astype(np.float32)

    def set_host_allreduce(self, fn):
        pass

    def invalidate_v(self):
        pass

    def update_w(self):
        oracle.nmf_update_w(self.V, self.W, self.H)

    def update_h(self):
        ps = dist.allreduce_sum_array(np.concatenate([self.W.T.dot(self.V), self.W.T.dot(self.W)], axis=1))
        P, S = ps[:, :self.n], ps[:, self.n:]
        self.H = (self.H * P) / (S.dot(self.H) + 1e-9)

    def frobenius(self):
        e2 = dist.allreduce_sum_array(np.array([np.sum((self.V - self.W.dot(self.H)) ** 2)]))
        return float(np.sqrt(e2[0]))

    def factorize(self, niter, compute_w=True, compute_h=True, compute_err=True, conv_eps=1e-8):
        ferr = np.zeros(max(niter, 1))
        for i in range(niter):
            if compute_w:
                self.update_w()
            if compute_h:
                self.update_h()
            if compute_err:
                ferr[i] = self.frobenius()
                if i > 1 and abs(ferr[i] - ferr[i - 1]) / self.n < conv_eps:      # nmf.py:134-139, 198-202
                    return ferr[:niter], i + 1, i
        return (ferr[:niter] if compute_err else None), niter, -1

    def close(self):
        pass


def gather_rows(block):
    parts = dist.allgather_bytes(np.ascontiguousarray(block, dtype=np.float64).tobytes())
    return np.concatenate([np.frombuffer(p, dtype=np.float64).reshape(-1, block.shape[1]) for p in parts], axis=0)


def check(name, got, want, tol):
    err = np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300)
    print("rank %d %s rel %.3g" % (dist.world().rank, name, err))
    assert err < tol, (name, err, tol)


def main():
    w = dist.init_from_env()
    assert w.size > 1
    if FAKE:
        _lib.Context = FakeContext
        pymf_amd.nmf._fingerprint = lambda a: (id(a), a.shape, float(np.sum(a)))     # no library needed
    tolx, tolf = (1e-6, 1e-6) if FAKE else (2e-5, 1e-5)   # the double hands W, H back as float32, like the library
    m, n, k = 1003, 256, 16                    # uneven split over the ranks
    V = np.random.RandomState(7).random_sample((m, n)).astype(np.float32)
    lo, hi = w.row_range(m)

    # ---- NMF, lazily initialised: ranks start with DIFFERENT seeds, rank 0's stream must decide ----
    np.random.seed(100 + w.rank)
    mdl = pymf_amd.NMF(V[lo:hi], num_bases=k)
    mdl.factorize(niter=4)
    assert mdl.W.shape == (hi - lo, k) and mdl.H.shape == (k, n)
    np.random.seed(100)
    ref = oracle.NMFOracle(V, num_bases=k)
    ref.factorize(niter=4)
    check("NMF W", gather_rows(mdl.W), ref.W, tolx)
    check("NMF H", mdl.H, ref.H, tolx)
    check("NMF ferr", mdl.ferr, ref.ferr, tolf)
    hs = dist.allgather_bytes(np.ascontiguousarray(mdl.H).tobytes())
    assert all(h == hs[0] for h in hs), "H must be bit-identical on every rank"
    digest("NMF W", mdl.W); digest("NMF H", mdl.H); digest("NMF ferr", mdl.ferr)
    if not FAKE:
        print("collective %s" % mdl._ctx.collective_name)
    # single hooks and the error under the multi-rank world
    mdl.update_w(); ref.update_w()
    mdl.update_h(); ref.update_h()
    check("NMF hooks W", gather_rows(mdl.W), ref.W, tolx)
    check("NMF hooks H", mdl.H, ref.H, tolx)
    assert abs(mdl.frobenius_norm() - ref.frobenius_norm()) <= tolf * ref.frobenius_norm()

    # ---- rank skew and the early exit (nmf.py:198-202): the last rank arrives late at every collective of the loop, the
    # data are exactly rank 1 so the fit becomes stationary within a few iterations -- every rank must leave the loop at
    # the SAME iteration (the decision is taken on all-reduced numbers only), none may be left waiting in a collective ----
    import time
    rs1 = np.random.RandomState(11)
    V1 = np.outer(rs1.random_sample(m) + 0.5, rs1.random_sample(n) + 0.5).astype(np.float32)
    if w.rank == w.size - 1:
        time.sleep(1.0)
    np.random.seed(300 + w.rank)
    e = pymf_amd.NMF(V1[lo:hi], num_bases=1)          # one basis on rank-1 data: exact after the first iteration
    e.factorize(niter=300)
    lens = [int(np.frombuffer(b, dtype=np.int64)[0]) for b in dist.allgather_bytes(np.array([len(e.ferr)], dtype=np.int64).tobytes())]
    print("rank %d early exit after %d of 300 iterations (all ranks: %s) ferr %s" % (w.rank, len(e.ferr), lens, e.ferr[:6]))
    assert len(set(lens)) == 1 and lens[0] < 300, lens
    fs = dist.allgather_bytes(np.ascontiguousarray(e.ferr, dtype=np.float64).tobytes())
    assert all(f == fs[0] for f in fs), "ferr must be bit-identical on every rank"
    digest("early-exit ferr", e.ferr)
    if FAKE:
        # ---- bench.py's sharded `parity_full_size` (round 6): rank 0 holds the unsharded float64 oracle's factors; every rank restarts
        # its shard from the seeded W0 rows / H0, runs the same iterations through the cross-rank sum and compares ITS row block
        # (scattered by rank 0); H digests are all-gathered.  Here on the NumPy stand-in of the device context: the collective
        # sequence (broadcasts, scatter, all-gathers with their tags) is what is under test ----
        import bench
        mb, nb_, kb = 517, 96, 8
        lob, hib = w.row_range(mb)
        fctx = FakeContext(_lib.ALGO_NMF, hib - lob, nb_, kb)
        fctx.set_v_dense(bench.gen_rows(np.random.RandomState(1234), mb, nb_, lob, hib))
        fac = None
        if w.rank == 0:
            Vb = bench.gen_rows(np.random.RandomState(1234), mb, nb_, 0, mb)
            np.random.seed(42)
            ob = oracle.NMFOracle(Vb, num_bases=kb)
            ob.W, ob.H = np.random.random((mb, kb)), np.random.random((kb, nb_))
            ob.W = ob.W.astype(np.float32).astype(np.float64)            # (bench.py uploads float32 rows of W0: gen_rows)
            ob.factorize(niter=6, compute_err=False)
            fac = {"iters": 6, "W": ob.W, "H": ob.H, "ferr": float(oracle.frobenius_norm(Vb, ob.W, ob.H))}
        par = bench.sharded_parity_nmf(fctx, dist, fac, mb, nb_, kb, lob, hib)
        if w.rank == 0:
            print("sharded parity: %s" % {k_: v_ for k_, v_ in par.items() if k_ != "against"})
            assert par["ranks"] == w.size and par["iters"] == 6 and len(par["relW_per_rank"]) == w.size
            assert par["h_identical_across_ranks"] is True and par["ferr_identical_across_ranks"] is True
            assert par["relW_max_over_ranks"] < 1e-6 and par["relH_max_over_ranks"] < 1e-6 and par["relferr"] < 1e-6, par
        else:
            assert par is None

    if not FAKE:
        # ---- loops that stop inside a chunk, back to back on the bare context (no per-call vote between them: bench.py's
        # pattern): the folded exchanges behind a stop push nothing, so their sequence numbers must be taken back or the two-slot
        # scheme of pmf_ipc.h loses its parity after an odd number of them (round-5 advisor).  W is disturbed between the calls so
        # that every loop runs a few iterations before it stops again -- at a different distance from its chunk's end each time;
        # the last rank lags, the others run ahead into the next call's exchanges ----
        ctx = e._ctx
        seen = []
        for rep in range(7):
            Wd = np.array(e.W, dtype=np.float64) * (1.0 + 0.3 * np.random.RandomState(50 + rep).random_sample((hi - lo, 1)))
            ctx.set_w(Wd)
            if w.rank == w.size - 1 and rep % 2 == 1:
                time.sleep(0.2)
            fe, done, conv = ctx.factorize(8 + rep)           # one ordinary iteration, then ONE chunk of 7 + rep: the stop leaves an
                                                              # odd number of skipped exchanges behind in every second call
            seen.append((int(done), int(conv)))
            digest("stopped loop %d ferr" % rep, np.asarray(fe[:done], dtype=np.float64))
        hh = ctx.get_h()
        hs = dist.allgather_bytes(np.ascontiguousarray(hh).tobytes())
        assert all(h == hs[0] for h in hs), "H must stay bit-identical on every rank across stopped loops"
        alls = dist.allgather_bytes(np.array(seen, dtype=np.int64).tobytes())
        assert all(a == alls[0] for a in alls), "every rank must stop at the same iterations"
        print("rank %d stopped loops (done, converged_at): %s" % (w.rank, seen))
        assert sum(1 for (d, cv), r in zip(seen, range(7)) if 0 <= cv and d < 8 + r) >= 4, seen     # ... most of them stop early
        digest("stopped loops H", hh)

    if not FAKE:
        # ---- SNMF (mixed-sign data) ----
        Vs = (V - 0.4).astype(np.float32)
        np.random.seed(200 + w.rank)
        s = pymf_amd.SNMF(Vs[lo:hi], num_bases=k)
        s.factorize(niter=3)
        np.random.seed(200)
        so = oracle.SNMFOracle(Vs, num_bases=k)
        so.factorize(niter=3)
        check("SNMF W", gather_rows(s.W), so.W, 5e-5)
        check("SNMF H", s.H, so.H, 2e-5)
        check("SNMF ferr", s.ferr, so.ferr, 2e-5)
        digest("SNMF W", s.W); digest("SNMF H", s.H)
        # ---- NMFALS (nmfals.py:70-97): the row QPs are local to a rank, the column QPs run on the all-reduced
        # (W^T V | W^T W) on every rank -- H must come out bit-identical everywhere ----
        np.random.seed(400 + w.rank)
        a = pymf_amd.NMFALS(V[lo:hi], num_bases=8)
        a.factorize(niter=3)
        np.random.seed(400)
        ao = oracle.NMFALSOracle(V, num_bases=8)
        ao.factorize(niter=3)
        check("NMFALS W", gather_rows(a.W), ao.W, 1e-4)
        check("NMFALS H", a.H, ao.H, 1e-4)
        check("NMFALS ferr", a.ferr, ao.ferr, 1e-5)
        ha = dist.allgather_bytes(np.ascontiguousarray(a.H).tobytes())
        assert all(h == ha[0] for h in ha), "NMFALS: H must be bit-identical on every rank"
        digest("NMFALS W", a.W); digest("NMFALS H", a.H)
        # ---- RNMF: init_h normalises the columns of W over ALL ranks' rows ----
        from pymf_amd.rnmf import RNMF
        Vr = V.copy()
        Vr.flat[np.random.RandomState(3).randint(0, Vr.size, size=Vr.size // 300)] += 5.0
        np.random.seed(300 + w.rank)
        r = RNMF(Vr[lo:hi], num_bases=k, lamb=1.0)
        r.factorize(niter=3)
        np.random.seed(300)
        ro = oracle.RNMFOracle(Vr, num_bases=k, lamb=1.0)
        ro.factorize(niter=3)
        check("RNMF W", gather_rows(r.W), ro.W, 2e-3)
        check("RNMF H", r.H, ro.H, 2e-3)
        check("RNMF ferr", r.ferr, ro.ferr, 2e-4)
        digest("RNMF W", r.W); digest("RNMF H", r.H)
        # ---- NNDSVD: Gram matrix and split norms summed over the ranks ----
        nd = pymf_amd.NNDSVD(V[lo:hi], num_bases=6)
        nd.factorize()
        no = oracle.NNDSVDOracle(V, num_bases=6)
        no.factorize()
        check("NNDSVD W", gather_rows(nd.W), no.W, 2e-3)
        check("NNDSVD H", nd.H, no.H, 2e-3)
    dist.barrier()
    dist.shutdown()
    print("rank %d ok" % w.rank)


if __name__ == "__main__":
    main()
