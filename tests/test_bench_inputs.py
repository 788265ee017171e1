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
