"""bench.py input protocol on CPU: the per-rank chunked generator must
reproduce exactly the rows of the single-process legacy-RNG streams, for any row partition."""
import os

import numpy as np
import pytest

import bench
from conftest import ROOT
from pymf_amd.dist import shard_rows


def test_gen_rows_matches_unsharded_stream():
    m, n, k = 1000, 7, 3
    V_full = np.random.RandomState(1234).random_sample((m, n)).astype(np.float32)
    np.random.seed(42)
    W_full = np.random.random((m, k))
    H_full = np.random.random((k, n))
    for size in (1, 2, 3, 8):
        for rank in range(size):
            lo, hi = shard_rows(m, rank, size)
            V = bench.gen_rows(np.random.RandomState(1234), m, n, lo, hi, chunk=64)
            np.testing.assert_array_equal(V, V_full[lo:hi])
            np.random.seed(42)
            W = bench.gen_rows(np.random.mtrand._rand, m, k, lo, hi, chunk=64)
            H = np.random.random((k, n))          # drawn after ALL of W, as nmf.py:173-177 does
            np.testing.assert_array_equal(W, W_full[lo:hi].astype(np.float32))
            np.testing.assert_array_equal(H, H_full)


def test_cpu_baseline_shape():
    for cfg, (m, n, k) in (("cfg4", (4096, 64, 8)), ("cfg3", (64, 48, 6)), ("cfg5", (2048, 64, 16))):
        out = bench.cpu_baseline(cfg, m, n, k, budget_s=0.5)
        assert out["kind"] == "port" and out["unit"] == "iter/s" and out["value"] > 0 and out["cores"] >= 1
        assert "sample" in out


def test_gen_csr_row_blocks_partition_the_matrix():
    """The per-rank CSR generator (fast stand-in): row blocks of any partition concatenate to the whole."""
    m, n = 5000, 32
    ip, ix, vv = bench.gen_csr(m, n, 0.05, 0, m, True)
    assert ip[0] == 0 and ip[-1] == len(vv) == len(ix) and len(ip) == m + 1
    for size in (2, 3):
        cols, vals, rows = [], [], 0
        for rank in range(size):
            lo, hi = shard_rows(m, rank, size)
            a, b, c = bench.gen_csr(m, n, 0.05, lo, hi, True)
            assert a[0] == 0 and len(a) == hi - lo + 1
            np.testing.assert_array_equal(np.diff(a), np.diff(ip[lo:hi + 1]))
            cols.append(b); vals.append(c); rows += hi - lo
        np.testing.assert_array_equal(np.concatenate(cols), ix)
        np.testing.assert_array_equal(np.concatenate(vals), vv)
        assert rows == m


@pytest.mark.gpu
def test_two_rank_bench_line_on_one_gpu():
    """The N > 1 harness end to end on a 1-GPU box: `python -m torch.distributed.run --nproc-per-node 2 bench.py
    --gpus 2 --debug-share-gpu` (both ranks on device 0, each with its own 1-rank RCCL communicator: rows still
    sharded, barriers and the max over ranks still used).  Rank 0 prints ONE line; both ranks' own step times are
    in it and agree (same work, same GPU), and `value` is K over the slowest."""
    import json
    import socket
    import subprocess
    import sys
    def free_port():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        return port

    errs = []
    for attempt in range(3):        # the rendezvous port is picked, released and re-bound by the launcher: a rare race
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "2",
               "--rows", "131072", "--fill", "device", "--debug-share-gpu", "--no-cpu-baseline", "--preroll-ms", "20"]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900,
                           env=dict(os.environ, OMP_NUM_THREADS="4"))
        if p.returncode == 0:
            break
        errs.append(p.stderr.decode()[-1500:])
        print("attempt %d of the two-rank launch failed:\n%s" % (attempt, errs[-1]))
    assert p.returncode == 0, "\n----\n".join(errs)
    lines = [json.loads(l) for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    j = lines[0]
    assert j["n_gpus"] == 2 and j["steps"] == 20 and len(j["rank_ms_per_step"]) == 2
    a, b = j["rank_ms_per_step"]
    assert abs(a - b) <= 0.5 * max(a, b), j["rank_ms_per_step"]       # two processes time-slicing one GPU
    assert abs(j["ms_per_step"] - max(a, b)) < 1e-9 and abs(j["value"] - 1e3 / max(a, b)) < 1e-6 * j["value"]
    assert j["roofline"]["traffic_source"] and j["config"]["class_factorize"] is None
