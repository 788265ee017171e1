"""world_size-N worker (CPU, socket rendezvous of pymf_amd.dist): the rendezvous plumbing of pymf_amd.dist and the
row-sharded formulation of one NMF iteration: rows of V/W are independent
given H; update_h needs only the SUM over ranks of (W_r^T V_r | W_r^T W_r)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pymf_amd import dist          # noqa: E402
from oracle import NMFOracle, nmf_update_w      # noqa: E402


def main():
    fake_id = bytes((7 * i + 3) % 251 for i in range(128))
    w = dist.init_from_env(make_nccl_id=lambda: fake_id)
    if "--mismatch" in sys.argv:
        # rank 0 alone makes a collective call (the "rank-0-only checkpoint" pattern); the peers' NEXT, unrelated collective
        # meets it: every rank must get CollectiveMismatch instead of mixed payloads or a hang
        try:
            if w.rank == 0:
                dist.allreduce_sum_array(np.ones(3))
            dist.allreduce_max(1.0)
            print("rank %d: no error" % w.rank)
        except dist.CollectiveMismatch as e:
            assert "disagree" in str(e), str(e)
            print("rank %d ok (mismatch caught)" % w.rank)
        return
    if "--watchdog" in sys.argv:
        import time
        assert dist.same_node()
        with dist.Watchdog("a step that returns in time", 30.0):
            pass
        dist.barrier()
        with dist.Watchdog("the step under test", 1.0):
            if w.rank == 1:
                time.sleep(30.0)              # stands for a rank stuck in ncclCommInitRank / an IPC wait
        print("rank %d ok" % w.rank)
        if w.rank == 0:
            time.sleep(3.0)                   # (outlives the peer's time-box)
        return
    assert w.size == int(os.environ["WORLD_SIZE"]) and w.rank == int(os.environ["RANK"])
    assert w.nccl_id == fake_id, "rank 0's unique id must reach every rank unchanged"
    m, n, k = 203, 40, 6
    lo, hi = w.row_range(m)
    covered = dist.allreduce_sum_array(np.array([hi - lo], dtype=np.float64))
    assert int(covered[0]) == m
    V = np.random.RandomState(1).random_sample((m, n))
    np.random.seed(2)
    W0, H0 = np.random.random((m, k)), np.random.random((k, n))
    # unsharded reference iteration
    ref = NMFOracle(V, num_bases=k)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=1, compute_err=False)
    # sharded: local update_w, all-reduce of the partials, identical H step on every rank
    Vr, Wr, H = V[lo:hi], W0[lo:hi].copy(), H0.copy()
    nmf_update_w(Vr, Wr, H)
    ps = np.concatenate([Wr.T.dot(Vr), Wr.T.dot(Wr)], axis=1)
    ps = dist.allreduce_sum_array(ps)
    P, S = ps[:, :n], ps[:, n:]
    H = (H * P) / (S.dot(H) + 1e-9)
    assert np.allclose(Wr, ref.W[lo:hi], rtol=1e-12, atol=0)
    assert np.allclose(H, ref.H, rtol=1e-10, atol=0)
    # H must be bit-identical on all ranks (same reduced inputs + same arithmetic)
    hsum = dist.allreduce_sum_array(H.copy())
    assert np.array_equal(hsum, H * w.size) or np.allclose(hsum, H * w.size, rtol=1e-15)
    dist.barrier()
    t = dist.allreduce_max(float(w.rank + 1))
    assert t == float(w.size)
    # the class-level plumbing: row counts -> global span, rank 0's RNG stream on every rank
    counts = dist.allgather_int(hi - lo)
    assert sum(counts) == m and counts[w.rank] == hi - lo
    np.random.seed(1000 + w.rank)                      # ranks start with DIFFERENT streams
    dist.share_rng_state()
    from pymf_amd.nmf import _draw_rows
    Wl = _draw_rows(m, k, lo, hi)
    Hl = np.random.random((k, n))
    np.random.seed(1000)
    assert np.array_equal(Wl, np.random.random((m, k))[lo:hi])
    assert np.array_equal(Hl, np.random.random((k, n)))
    got = dist.broadcast_array(np.arange(6, dtype=np.float32).reshape(2, 3) * (w.rank + 1), src=w.size - 1)
    assert np.array_equal(got, np.arange(6, dtype=np.float32).reshape(2, 3) * w.size)
    # round 6 (bench.py's sharded parity): rank 0 hands every rank ITS block, and collects one block per rank, without the
    # payloads travelling to everybody -- same frame check (sequence number + tag) as every other collective
    blocks = [np.full((r + 2, 3), float(r + 1)) for r in range(w.size)]
    mine = np.frombuffer(dist.scatter_bytes([b.tobytes() for b in blocks] if w.rank == 0 else None, tag="blk"), dtype=np.float64)
    assert mine.shape == ((w.rank + 2) * 3,) and np.all(mine == float(w.rank + 1))
    back = dist.gather_bytes((mine * 2.0).tobytes(), tag="blk2")
    if w.rank == 0:
        assert [np.frombuffer(b, dtype=np.float64)[0] for b in back] == [2.0 * (r + 1) for r in range(w.size)]
        assert [len(b) for b in back] == [8 * 3 * (r + 2) for r in range(w.size)]
    else:
        assert back is None
    try:                                   # a scatter on rank 0 against a gather elsewhere: different tags -> every rank fails loudly
        if w.rank == 0:
            dist.scatter_bytes([b"x"] * w.size, tag="one")
        else:
            dist.gather_bytes(b"y", tag="other")
        raise AssertionError("mismatched collectives went through")
    except dist.CollectiveMismatch:
        pass
    if "--assert-no-torch" in sys.argv:
        assert "torch" not in sys.modules, "the product must not import torch"
    dist.shutdown()
    print("rank %d ok" % w.rank)


if __name__ == "__main__":
    main()
