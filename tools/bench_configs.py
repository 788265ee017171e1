"""Time the secondary BASELINE configs (cfg2, cfg3, cfg5) through the C ABI on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pymf_amd import _lib


def run(name, algo, m, n, k, niter, csr_density=None, compute_err=False):
    ctx = _lib.Context(algo, m, n, k)
    if csr_density is None:
        ctx.fill_v_uniform(1234)
    else:
        import scipy.sparse as sp
        t = time.time()
        rs = np.random.RandomState(1234)
        nnz_row = rs.poisson(csr_density * n, size=m).astype(np.int64)
        nnz_row = np.minimum(nnz_row, n)
        indptr = np.concatenate([[0], np.cumsum(nnz_row)])
        nnz = int(indptr[-1])
        indices = rs.randint(0, n, size=nnz).astype(np.int32)     # duplicates are summed by the kernels
        vals = rs.random_sample(nnz).astype(np.float32)
        ctx.set_v_csr(indptr, indices, vals)
        print("  csr built: nnz=%d (%.2f/row) in %.1fs" % (nnz, nnz / m, time.time() - t))
    ctx.fill_w_uniform(42)
    ctx.fill_h_uniform(43)
    ctx.factorize(2, compute_err=False)
    t = time.time()
    ctx.factorize(niter, compute_err=compute_err)
    dt = time.time() - t
    print("%-40s path=%-18s %8.3f ms/iter  %9.2f it/s" % (name, ctx.path_name, dt / niter * 1e3, niter / dt), flush=True)
    ctx.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg2", "cfg3", "cfg5"]
    if "cfg2" in which:
        run("cfg2 NMF 65536x512 k=32", _lib.ALGO_NMF, 65536, 512, 32, 200)
    if "cfg3" in which:
        run("cfg3 NMFALS 262144x1024 k=64", _lib.ALGO_NMFALS, 262144, 1024, 64, 3)
    if "cfg5shard" in which:
        run("cfg5 SNMF CSR 524288x128 k=128 (1/8 shard)", _lib.ALGO_SNMF, 524288, 128, 128, 20, csr_density=0.01)
    if "cfg5" in which:
        run("cfg5 SNMF CSR 524288x128 k=128 (1/8 shard)", _lib.ALGO_SNMF, 524288, 128, 128, 20, csr_density=0.01)
        run("cfg5 SNMF CSR 4194304x128 k=128 (1 GPU)", _lib.ALGO_SNMF, 4194304, 128, 128, 10, csr_density=0.01)
    if "snmf_dense" in which:
        run("SNMF dense 1048576x256 k=64", _lib.ALGO_SNMF, 1048576, 256, 64, 20)
