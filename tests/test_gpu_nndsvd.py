"""GPU parity for NNDSVD (SURVEY 8(f) 'next' row 4) against goldens produced by the reference and
against the float64 oracle.  float32 tolerance: the device forms data^T data and data V S^-1 on
fp32 MFMA (the reference forms them in the data's dtype, float32 for these goldens), so the factors
agree to ~1e-4 relative Frobenius; stated bound 5e-4 on W and H, 1e-4 on ferr."""
import numpy as np
import pytest

from conftest import load_golden, rel_fro, close

pytestmark = pytest.mark.gpu

CASES = ["nndsvd_300x40_k6", "nndsvd_40x300_k6", "nndsvd_cfg1_k4", "nndsvd_37x29_k5", "nndsvd_doc_k2",
         "nndsvd_1024x256_k64"]


@pytest.fixture(scope="module")
def pm():
    import pymf_amd
    from pymf_amd import _lib
    assert _lib.device_count() >= 1
    return pymf_amd


@pytest.mark.parametrize("name", CASES)
def test_nndsvd_vs_reference_golden(pm, name):
    g = load_golden(name)
    mdl = pm.NNDSVD(g["V"], num_bases=int(g["k"]))
    mdl.factorize(niter=5, compute_h=False)              # arguments are overridden (nndsvd.py:111-114)
    assert mdl.W.shape == g["W"].shape and mdl.H.shape == g["H"].shape
    assert mdl.W.dtype == np.float64 and mdl.H.dtype == np.float64      # np.zeros init (nndsvd.py:70,73)
    assert (mdl.W >= 0).all() and (mdl.H >= 0).all()
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 3e-4, rel_fro(mdl.W, g["W"], what="mdl.W")
    assert rel_fro(mdl.H, g["H"], what="mdl.H") < 3e-4, rel_fro(mdl.H, g["H"], what="mdl.H")
    assert len(mdl.ferr) == 1
    close(mdl.ferr, g["ferr"], rtol=1e-4, atol=1e-5, what="mdl.ferr")


def test_nndsvd_then_nmf_matches_reference(pm):
    """The documented use (nndsvd.py:56-64): NMF started from the NNDSVD factors."""
    g = load_golden("nndsvd_300x40_k6")
    nd = pm.NNDSVD(g["V"], num_bases=int(g["k"]))
    nd.factorize()
    mdl = pm.NMF(g["V"], num_bases=int(g["k"]))
    mdl.W = nd.W
    mdl.H = nd.H
    mdl.factorize(niter=10)
    close(mdl.ferr, g["ferr_nmf10"], rtol=8e-7, what="mdl.ferr")
    assert rel_fro(mdl.W, g["W_nmf10"], what="mdl.W") < 6e-5 and rel_fro(mdl.H, g["H_nmf10"], what="mdl.H") < 6e-5
    # the head start the initialiser exists for: the first NMF error is below a random start's
    rnd = pm.NMF(g["V"], num_bases=int(g["k"]))
    np.random.seed(0)
    rnd.factorize(niter=1)
    assert mdl.ferr[0] < rnd.ferr[0]


@pytest.mark.parametrize("shape,k", [((4096, 256), 64), ((700, 130), 17), ((129, 1000), 9), ((200, 64), 64), ((64, 64), 16),
                                     ((5000, 3), 3), ((2, 2), 1),
                                     # min(m, n) > 1024 (svd.py:125-148 has no size limit): 750 and 1250 rotation
                                     # pairs per Jacobi step, more than one per thread of a workgroup
                                     ((3000, 1500), 12), ((1100, 2500), 8)])
def test_nndsvd_vs_float64_oracle(pm, shape, k):
    from oracle import nndsvd_closed_form
    V = np.random.RandomState(sum(shape) + k).random_sample(shape).astype(np.float32)
    mdl = pm.NNDSVD(V, num_bases=k)
    mdl.factorize()
    W, H = nndsvd_closed_form(V, k)
    # trailing singular directions of a random matrix are nearly degenerate: compare the product
    # basis by basis only where the gap allows, and the factors overall
    assert rel_fro(mdl.W, W, what="mdl.W") < 2e-5, rel_fro(mdl.W, W, what="mdl.W")
    assert rel_fro(mdl.H, H, what="mdl.H") < 2e-5, rel_fro(mdl.H, H, what="mdl.H")
    ref_err = np.linalg.norm(V.astype(np.float64) - W @ H)
    assert abs(mdl.ferr[0] - ref_err) <= 1e-4 * max(ref_err, 1e-3)


@pytest.mark.parametrize("kind,shape,k", [("uniform", (3000, 1500), 12), ("uniform", (4096, 2048), 200), ("binary", (5000, 2048), 48),
                                          ("dupcols", (3000, 1024), 20), ("lowrank_exact", (2500, 1280), 30)])
def test_nndsvd_topk_solver_equals_full_jacobi(pm, kind, shape, k):
    """pmf_nndsvd_init's two eigen-solvers -- all n pairs of data^T data by Jacobi (svd.py:125-131 takes them all from
    eigh and keeps the leading ones), or the k largest by the filtered subspace iteration of pmf_topk.h -- give the same
    W and H: dominant eigenvalue 4 000 x the bulk (uniform), sparse binary data, every eigenvalue double (duplicated
    columns), an exactly rank-30 matrix asked for 30 bases."""
    from pymf_amd import _lib
    rs = np.random.RandomState(sum(shape) + k)
    m, n = shape
    if kind == "uniform":
        V = rs.random_sample(shape)
    elif kind == "binary":
        V = (rs.random_sample(shape) < 0.05).astype(np.float64)
    elif kind == "dupcols":
        a = rs.random_sample((m, n // 2)); V = np.concatenate([a, a], axis=1)
    else:
        V = rs.random_sample((m, 30)) @ rs.random_sample((30, n))
    V = V.astype(np.float32)
    out = {}
    for topk in (1, 0):
        ctx = _lib.Context(_lib.ALGO_NMF, m, n, k)
        ctx.set_v_dense(V)
        ctx.set_option("nndsvd_topk", topk)
        assert ctx.nndsvd_init() == k
        out[topk] = (ctx.get_w(), ctx.get_h())
        ctx.close()
    # (a double eigenvalue leaves the basis of its eigenspace open: compare what NNDSVD makes of it, W H)
    if kind == "dupcols":
        a, b = out[1][0].astype(np.float64) @ out[1][1], out[0][0].astype(np.float64) @ out[0][1]
        assert rel_fro(a, b, what="W H, top-k vs Jacobi") < 1e-5
    else:
        tol = 2e-4 if kind == "lowrank_exact" else 1e-6     # rank 30 exactly: the 30th direction sits on float32 noise
        assert rel_fro(out[1][0], out[0][0], what="W, top-k vs Jacobi") < tol
        assert rel_fro(out[1][1], out[0][1], what="H, top-k vs Jacobi") < tol


def test_nndsvd_beyond_4096_columns_vs_float64_oracle(pm):
    """min(m, n) > 4096 (svd.py:125-148 has no size limit): only the top-k solver exists there."""
    from oracle import nndsvd_closed_form
    shape, k = (5000, 4608), 32
    V = np.random.RandomState(7).random_sample(shape).astype(np.float32)
    mdl = pm.NNDSVD(V, num_bases=k)
    mdl.factorize()
    W, H = nndsvd_closed_form(V, k)
    assert rel_fro(mdl.W, W, what="mdl.W") < 2e-5 and rel_fro(mdl.H, H, what="mdl.H") < 2e-5
    wide = pm.NNDSVD(np.ascontiguousarray(V.T[:, :4700]), num_bases=8)      # 4608 x 4700: the transposed problem, still > 4096
    wide.factorize()
    W2, H2 = nndsvd_closed_form(np.ascontiguousarray(V.T[:, :4700]), 8)
    assert rel_fro(wide.W, W2, what="wide.W") < 2e-5 and rel_fro(wide.H, H2, what="wide.H") < 2e-5


def test_nndsvd_rank_deficient_raises(pm):
    from pymf_amd._lib import PmfError
    V = np.outer(np.arange(1, 41, dtype=np.float32), np.arange(1, 9, dtype=np.float32))   # rank 1
    mdl = pm.NNDSVD(V, num_bases=3)
    with pytest.raises(PmfError):                        # reference: IndexError at nndsvd.py:94
        mdl.factorize()
    # ... and through the top-k solver (n > 1024): rank 3, 40 bases asked for (0/1 factors: the float32 Gram matrix is exact)
    rs = np.random.RandomState(3)
    V = ((rs.random_sample((2000, 3)) < 0.5).astype(np.float32) @ (rs.random_sample((3, 1100)) < 0.5).astype(np.float32))
    mdl = pm.NNDSVD(V, num_bases=40)
    with pytest.raises(PmfError):
        mdl.factorize()


def test_nndsvd_init_through_the_c_abi(pm):
    """pmf_nndsvd_init on an NMF context, then the hot loop continues from the device-resident factors."""
    from pymf_amd import _lib
    from oracle import nndsvd_closed_form, NMFOracle
    V = np.random.RandomState(77).random_sample((2048, 192)).astype(np.float32)
    ctx = _lib.Context(_lib.ALGO_NMF, 2048, 192, 24)
    ctx.set_v_dense(V)
    assert ctx.nndsvd_init() == 24
    W0, H0 = ctx.get_w(), ctx.get_h()
    Wr, Hr = nndsvd_closed_form(V, 24)
    assert rel_fro(W0, Wr, what="W0") < 6e-6 and rel_fro(H0, Hr, what="H0") < 6e-6
    ferr, done, _ = ctx.factorize(5, True, True, True)
    o = NMFOracle(V, num_bases=24)
    o.W, o.H = W0.astype(np.float64), H0.astype(np.float64)
    o.factorize(niter=5)
    close(ferr[:done], o.ferr, rtol=3e-7, what="ferr[:done]")
    ctx.close()


def test_nndsvd_single_rank_communicator(pm):
    """The row-sharded form sums the Gram matrix and the split norms with ncclAllReduce(double):
    through a 1-rank RCCL communicator the result must equal the communicator-free one bit for bit."""
    from pymf_amd import _lib
    V = np.random.RandomState(21).random_sample((3000, 128)).astype(np.float32)
    outs = []
    for nid in (None, _lib.nccl_unique_id()):
        ctx = _lib.Context(_lib.ALGO_NMF, 3000, 128, 12, device=0, rank=0, nranks=1, nccl_id=nid)
        ctx.set_v_dense(V)
        assert ctx.nndsvd_init() == 12
        outs.append((ctx.get_w(), ctx.get_h()))
        ctx.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
