"""bench.py input protocol on CPU: the per-rank chunked generator must
reproduce exactly the rows of the single-process legacy-RNG streams, for any row partition."""
import numpy as np

import bench
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
    out = bench.cpu_baseline(4096, 64, 8, budget_s=0.5)
    assert out["kind"] == "port" and out["unit"] == "iter/s" and out["value"] > 0 and out["cores"] >= 1
    assert "sample" in out
