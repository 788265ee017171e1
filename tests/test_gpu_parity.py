"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle and the reference's goldens.

Tolerances (float32 device arithmetic vs the float64-default reference):
  ||X_gpu - X_ref||_F / ||X_ref||_F <= 2e-5 for X in {W, H};  |ferr_gpu - ferr_ref|/ferr_ref <= 1e-5.
"""
import os

import numpy as np
import pytest

from conftest import load_golden, rel_fro, close

pytestmark = pytest.mark.gpu

TOL_X = 2e-5
TOL_F = 1e-5


@pytest.fixture(scope="module")
def pm():
    import pymf_amd
    from pymf_amd import _lib
    assert _lib.device_count() >= 1, "no HIP device: the GPU tests need an MI355X"
    return pymf_amd


def _run(cls, V, k, niter, W0, H0, **flags):
    mdl = cls(V, num_bases=k)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    mdl.factorize(niter=niter, **flags)
    return mdl


NMF_GOLD = ["nmf_cfg1_f64", "nmf_cfg1_f32", "nmf_512x128_k16", "nmf_cfg4s", "nmf_cfg2s",
            "nmf_cfg5s_dense", "nmf_37x29_k5", "nmf_reftest"]


@pytest.mark.parametrize("name", NMF_GOLD)
def test_nmf_vs_reference_golden(pm, name):
    g = load_golden(name)
    mdl = _run(pm.NMF, g["V"], int(g["k"]), int(g["niter"]), g["W0"], g["H0"])
    assert mdl.W.dtype == g["W"].dtype and mdl.H.dtype == g["H"].dtype
    assert len(mdl.ferr) == len(g["ferr"])
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 3e-6
    assert rel_fro(mdl.H, g["H"], what="mdl.H") < 4e-6
    close(mdl.ferr, g["ferr"], rtol=TOL_F, what="mdl.ferr")


SNMF_GOLD = ["snmf_cfg1_f64", "snmf_cfg1_f32", "snmf_512x128_k16", "snmf_cfg4s", "snmf_cfg2s",
             "snmf_sparse1pct", "snmf_37x29_k5", "snmf_reftest"]


@pytest.mark.parametrize("name", SNMF_GOLD)
def test_snmf_vs_reference_golden(pm, name):
    g = load_golden(name)
    W_before = g["W0"].copy()
    mdl = pm.SNMF(g["V"], num_bases=int(g["k"]))
    mdl.W, mdl.H = W_before, g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert mdl.W is not W_before                     # snmf.py:70 rebinds self.W
    assert len(mdl.ferr) == len(g["ferr"])
    # W crosses zero: Frobenius-relative only
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 5e-5
    assert rel_fro(mdl.H, g["H"], what="mdl.H") < 7e-6
    close(mdl.ferr, g["ferr"], rtol=5e-6, what="mdl.ferr")


@pytest.mark.parametrize("name,sparse", [("snmf_cfg5s_dense_f64", False), ("snmf_csr_k128_f64", False),
                                         ("snmf_csr_k128_f64", True)])
def test_snmf_cfg5_shape_class_vs_reference_golden(pm, name, sparse):
    """cfg5's shape class, k = n = 128: cond(H H^T) ~ 1e7.  Against the float64-default reference
    (SNMF on V.toarray() for the sparse case) at the STATED tolerances 5e-5 (W) / 2e-5 (H) -- the
    reference's own all-float32 run (W32 / H32 in the fixture) misses W by percents here; the device
    path holds because M^T = inv(H H^T) H is formed in float64 before the big product (k_snmf_mt).
    The fit is exact (k = n), so the error is rounding noise: compared on the scale of ||V||."""
    import scipy.sparse as sp
    g = load_golden(name)
    V = sp.csr_matrix(g["V"]) if sparse else g["V"]
    mdl = pm.SNMF(V, num_bases=int(g["k"]))
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]), compute_err=not sparse)
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 3e-5
    assert rel_fro(mdl.H, g["H"], what="mdl.H") < 3e-7
    # on record next to them: what the reference's own float32 arithmetic achieves on the same inputs
    assert rel_fro(g["W32"], g["W"], what="reference float32 path W (not the device)") < 4e-1
    assert rel_fro(g["H32"], g["H"], what="reference float32 path H (not the device)") < 9e-5
    assert rel_fro(mdl.W, g["W"]) < 1e-2 * rel_fro(g["W32"], g["W"])
    if not sparse:
        assert len(mdl.ferr) == len(g["ferr"])
        vn = float(np.linalg.norm(g["V"]))
        close(mdl.ferr / vn, g["ferr"] / vn, rtol=1.0, atol=2e-5, what="mdl.ferr / ||V|| (exact fit)")
        assert np.all(mdl.ferr < g["ferr32"])            # and closer to the exact fit than the reference's float32 run


@pytest.mark.parametrize("m,n,k", [(64, 64, 16), (100, 70, 3), (257, 130, 33), (1000, 256, 64),
                                   (513, 320, 100), (4096, 256, 64), (130, 1100, 20)])
def test_nmf_vs_oracle_shapes(pm, m, n, k):
    from oracle import NMFOracle
    rs = np.random.RandomState(m + n + k)
    V = rs.random_sample((m, n)).astype(np.float32)
    W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
    mdl = _run(pm.NMF, V, k, 6, W0, H0)
    ref = NMFOracle(V, num_bases=k)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=6)
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 4e-6
    assert rel_fro(mdl.H, ref.H, what="mdl.H") < 9e-7
    close(mdl.ferr, ref.ferr, rtol=3e-7, what="mdl.ferr")


def test_single_hooks_match_oracle(pm):
    """update_w() / update_h() / frobenius_norm() callable singly, one C call each."""
    from oracle import nmf_update_w, nmf_update_h, frobenius_norm
    rs = np.random.RandomState(11)
    V = rs.random_sample((333, 200)).astype(np.float32)
    W, H = rs.random_sample((333, 24)), rs.random_sample((24, 200))
    mdl = pm.NMF(V, num_bases=24)
    mdl.W, mdl.H = W.copy(), H.copy()
    assert abs(mdl.frobenius_norm() - frobenius_norm(V, W, H)) / frobenius_norm(V, W, H) < TOL_F
    mdl.update_w()
    nmf_update_w(V, W, H)
    assert rel_fro(mdl.W, W, what="mdl.W") < 7e-7
    mdl.update_h()
    nmf_update_h(V, W, H)
    assert rel_fro(mdl.H, H, what="mdl.H") < 2e-7
    assert abs(mdl.frobenius_norm() - frobenius_norm(V, W, H)) / frobenius_norm(V, W, H) < TOL_F


def test_flag_sequence_and_resume(pm):
    """tests/test_pymf.py:92-95 -- repeated factorize() with flags; golden from the reference."""
    g = load_golden("nmf_flagseq")
    np.random.seed(int(g["seed"]))
    mdl = pm.NMF(g["V"], num_bases=int(g["k"]))
    mdl.factorize(niter=5)
    assert rel_fro(mdl.W, g["W_a"], what="mdl.W") < 6e-7 and rel_fro(mdl.H, g["H_a"], what="mdl.H") < 5e-7
    close(mdl.ferr, g["ferr_a"], rtol=3e-7, what="mdl.ferr")
    mdl.factorize(niter=5, compute_h=False)
    assert rel_fro(mdl.W, g["W_b"], what="mdl.W") < 2e-6 and rel_fro(mdl.H, g["H_b"], what="mdl.H") < 5e-7
    mdl.factorize(niter=5, compute_w=False)
    assert rel_fro(mdl.W, g["W_c"], what="mdl.W") < 2e-6 and rel_fro(mdl.H, g["H_c"], what="mdl.H") < 7e-7
    before = mdl.ferr.copy()
    mdl.factorize(niter=5, compute_err=False)
    assert rel_fro(mdl.W, g["W_d"], what="mdl.W") < 2e-6 and rel_fro(mdl.H, g["H_d"], what="mdl.H") < 8e-7
    np.testing.assert_array_equal(mdl.ferr, before)          # nmf.py:179-180
    close(mdl.ferr, g["ferr_d"], rtol=1e-7, what="mdl.ferr")


def test_early_exit_truncates_ferr(pm):
    """nmf.py:198-202 -- exact data, compute_w=False: reference stops with len(ferr)==2."""
    g = load_golden("nmf_earlyexit")
    mdl = pm.NMF(g["V"], num_bases=2)
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=20, compute_w=False)
    assert len(mdl.ferr) == len(g["ferr"]) == 2
    close(mdl.H, g["H"], rtol=1e-5, atol=1e-6, what="mdl.H")


def test_sentinel_and_errors(pm):
    mdl = pm.NMF(np.ones((5, 4), dtype=np.float32), num_bases=2)
    assert mdl.frobenius_norm() == -123456                   # nmf.py:112
    with pytest.raises(TypeError):
        pm.NMF(np.ones((5, 4)), num_bases=2, niter=3)        # stale docstring ctor
    mdl.W = np.ones((5, 2), dtype=np.int64)
    mdl.H = np.ones((2, 4))
    with pytest.raises(TypeError):
        mdl.factorize(niter=1)                               # integer W fails in the reference too


def test_user_assigned_factors_are_reuploaded(pm):
    from oracle import NMFOracle
    rs = np.random.RandomState(2)
    V = rs.random_sample((80, 64)).astype(np.float32)
    mdl = pm.NMF(V, num_bases=8)
    np.random.seed(1)
    mdl.factorize(niter=2)
    W1, H1 = rs.random_sample((80, 8)), rs.random_sample((8, 64))
    mdl.W, mdl.H = W1.copy(), H1.copy()
    mdl.factorize(niter=3)
    ref = NMFOracle(V, num_bases=8)
    ref.W, ref.H = W1.copy(), H1.copy()
    ref.factorize(niter=3)
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 7e-7 and rel_fro(mdl.H, ref.H, what="mdl.H") < 4e-7
    mdl.H *= 0.5                                             # in-place edit must be noticed
    ref.H *= 0.5
    mdl.factorize(niter=1)
    ref.factorize(niter=1)
    assert rel_fro(mdl.H, ref.H, what="mdl.H") < 5e-7


def test_linearity_property_large(pm):
    """Size-independent property at a larger size: one update_w is homogeneous of
    degree 0 in H scaling pairs: W(V, W0, c*H) * c == W(V, W0, H) (up to the 1e-9 eps)."""
    rs = np.random.RandomState(4)
    m, n, k = 65536, 256, 64
    V = rs.random_sample((m, n)).astype(np.float32)
    W0, H0 = rs.random_sample((m, k)).astype(np.float32), rs.random_sample((k, n)).astype(np.float32)
    a = pm.NMF(V, num_bases=k)
    a.W, a.H = W0.copy(), H0.copy()
    a.update_w()
    b = pm.NMF(V, num_bases=k)
    b.W, b.H = W0.copy(), (2.0 * H0)
    b.update_w()
    assert rel_fro(2.0 * b.W, a.W, what="2.0 * b.W") < 1e-9


def test_rccl_path_single_rank_communicator(pm):
    """The per-iteration ncclAllReduce of (W^T V | W^T W) and of the residual scalar, run through
    a 1-rank RCCL communicator: results must equal the communicator-free context bit for bit."""
    from pymf_amd import _lib
    rs = np.random.RandomState(8)
    m, n, k = 4096, 256, 64
    V = rs.random_sample((m, n)).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = rs.random_sample((k, n)).astype(np.float32)
    outs = []
    for nid in (None, _lib.nccl_unique_id()):
        ctx = _lib.Context(_lib.ALGO_NMF, m, n, k, device=0, rank=0, nranks=1, nccl_id=nid)
        ctx.set_v_dense(V)
        ctx.set_w(W0)
        ctx.set_h(H0)
        ferr, done, conv = ctx.factorize(4)
        outs.append((ctx.get_w(), ctx.get_h(), ferr))
        ctx.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][2], outs[1][2])


def test_fused_and_tiled_paths_agree(pm):
    """The one-pass fused kernel and the two-pass tiled kernels implement the same iteration."""
    from pymf_amd import _lib
    rs = np.random.RandomState(12)
    m, n, k = 8192, 256, 64
    V = rs.random_sample((m, n)).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = rs.random_sample((k, n)).astype(np.float32)
    a = _lib.Context(_lib.ALGO_NMF, m, n, k)
    assert a.path_name.startswith("k_nmf_fused")
    a.set_v_dense(V); a.set_w(W0); a.set_h(H0)
    a.factorize(3, compute_err=False)               # fused
    b = _lib.Context(_lib.ALGO_NMF, m, n, k)
    b.set_option("force_tiled", 1)                   # k_rowgemm<EPI_NMF_W> + k_colgemm: the any-shape two-pass kernels
    assert b.path_name.startswith("tiled")
    b.set_v_dense(V); b.set_w(W0); b.set_h(H0)
    for _ in range(3):
        b.update_w()
        b.update_h()
    assert rel_fro(a.get_w(), b.get_w(), what="fused vs forced-tiled W") < 1e-6
    assert rel_fro(a.get_h(), b.get_h(), what="fused vs forced-tiled H") < 1e-6
    c = _lib.Context(_lib.ALGO_NMF, m, n, k)         # the single hooks on a fused shape run the one-pass kernel too:
    c.set_v_dense(V); c.set_w(W0); c.set_h(H0)       # hook by hook == one call, bit for bit
    for _ in range(3):
        c.update_w()
        c.update_h()
    np.testing.assert_array_equal(a.get_w(), c.get_w())
    np.testing.assert_array_equal(a.get_h(), c.get_h())


def test_full_size_properties_cfg4(pm):
    """BASELINE cfg4 size (1,048,576 x 256, k = 64), size-independent properties only:
    (1) the Lee-Seung objective never increases under the multiplicative updates,
    (2) the fused one-pass kernel and the two-pass tiled kernels agree on the same inputs,
    (3) the trace-identity residual equals the direct residual pass,
    (4) W, H stay non-negative and finite."""
    from pymf_amd import _lib
    m, n, k = 1048576, 256, 64
    a = _lib.Context(_lib.ALGO_NMF, m, n, k)
    a.fill_v_uniform(1234); a.fill_w_uniform(42); a.fill_h_uniform(43)
    assert a.path_name == "k_nmf_fused<4,4>"
    ferr, done, conv = a.factorize(6, compute_err=True)          # fused + trace identity
    assert done == 6 and conv < 0
    assert np.all(np.diff(ferr) <= 1e-6 * ferr[0]), ferr          # (1)
    b = _lib.Context(_lib.ALGO_NMF, m, n, k)
    b.set_option("force_tiled", 1)                                # k_rowgemm<EPI_NMF_W> + k_colgemm at the full size
    b.fill_v_uniform(1234); b.fill_w_uniform(42); b.fill_h_uniform(43)
    fb = []
    for _ in range(6):                                            # hooks: tiled kernels + direct residual
        b.update_w()
        b.update_h()
        b.set_w(b.get_w())                                        # "new" W invalidates (W^T V | W^T W): forces the direct pass
        fb.append(b.frobenius())
    close(ferr, np.array(fb), rtol=5e-8, what="cfg4 ferr: fused + trace identity vs tiled + direct residual")     # (2) + (3)
    Ha, Hb = a.get_h(), b.get_h()                                 # H depends on every row of W
    assert rel_fro(Ha, Hb, what="cfg4 H fused vs forced-tiled") < 2e-6
    assert np.isfinite(Ha).all() and Ha.min() >= 0                # (4)
    Wa = a.get_w()
    assert float(Wa.min()) >= 0.0 and np.isfinite(float(Wa.sum(dtype=np.float64)))
    Wb = b.get_w()
    sl = slice(0, m, 4097)
    assert rel_fro(Wa[sl], Wb[sl], what="cfg4 W[sl] fused vs forced-tiled") < 2e-6
    a.close(); b.close()


@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (1, 7, 2), (5, 1, 3), (3, 50, 4), (17, 9, 12), (64, 64, 64),
                                   (65, 257, 65), (33, 300, 128)])
def test_ragged_and_tiny_shapes(pm, m, n, k):
    """Edge shapes: single row/column, k > n, k > m, sizes straddling every padding boundary."""
    from oracle import NMFOracle
    rs = np.random.RandomState(100 + m + n + k)
    V = rs.random_sample((m, n)).astype(np.float32) + 0.05
    W0, H0 = rs.random_sample((m, k)) + 0.05, rs.random_sample((k, n)) + 0.05
    mdl = _run(pm.NMF, V, k, 4, W0, H0)
    ref = NMFOracle(V, num_bases=k)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=4)
    assert mdl.W.shape == (m, k) and mdl.H.shape == (k, n)
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 2e-6 and rel_fro(mdl.H, ref.H, what="mdl.H") < 8e-7
    if ref.ferr[-1] > 1e-5 * np.linalg.norm(V):
        close(mdl.ferr, ref.ferr, rtol=2e-4, atol=1e-6, what="mdl.ferr")
    else:
        # rank(V) <= k: the fit is exact and the residual is pure rounding noise (float32 here,
        # float64 in the reference), so neither its digits nor the early-exit iteration are comparable
        assert mdl.ferr[-1] < 1e-6 * max(1.0, np.linalg.norm(V))


def test_zero_rows_and_columns(pm):
    """All-zero rows/columns of V drive the matching W rows / H columns to exactly 0 (0*x/(d+1e-9))."""
    from oracle import NMFOracle
    rs = np.random.RandomState(5)
    V = rs.random_sample((70, 40)).astype(np.float32)
    V[10:20, :] = 0.0
    V[:, 5] = 0.0
    W0, H0 = rs.random_sample((70, 6)), rs.random_sample((6, 40))
    mdl = _run(pm.NMF, V, 6, 5, W0, H0)
    ref = NMFOracle(V, num_bases=6)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=5)
    assert np.all(mdl.W[10:20] == 0.0) and np.all(mdl.H[:, 5] == 0.0)
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 6e-7 and rel_fro(mdl.H, ref.H, what="mdl.H") < 5e-7


def test_snmf_fused_and_tiled_paths_agree(pm):
    """SNMF: one-pass fused kernel (factorize) vs the two-pass hooks on mixed-sign data."""
    from pymf_amd import _lib
    rs = np.random.RandomState(21)
    m, n, k = 8192, 256, 48
    V = (rs.random_sample((m, n)) - 0.3).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = (rs.random_sample((k, n)) + 0.1).astype(np.float32)
    a = _lib.Context(_lib.ALGO_SNMF, m, n, k)
    assert a.path_name.startswith("k_nmf_fused") and "snmf" in a.path_name
    a.set_v_dense(V); a.set_w(W0); a.set_h(H0)
    a.factorize(3, compute_err=False)
    b = _lib.Context(_lib.ALGO_SNMF, m, n, k)
    b.set_option("force_tiled", 1)
    b.set_option("snmf_gram", 0)
    b.set_v_dense(V); b.set_w(W0); b.set_h(H0)
    for _ in range(3):
        b.update_w()
        b.update_h()
    assert rel_fro(a.get_w(), b.get_w(), what="a.get_w()") < 2e-6
    assert rel_fro(a.get_h(), b.get_h(), what="a.get_h()") < 4e-7


def test_bitwise_reproducible(pm):
    """Fixed-order partial sums everywhere on the dense paths: two runs give identical bits."""
    from pymf_amd import _lib
    rs = np.random.RandomState(77)
    m, n, k = 20000, 256, 64
    V = rs.random_sample((m, n)).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = rs.random_sample((k, n)).astype(np.float32)
    outs = []
    for _ in range(2):
        for algo in (_lib.ALGO_NMF, _lib.ALGO_SNMF, _lib.ALGO_BNMF):
            ctx = _lib.Context(algo, m, n, k)
            ctx.set_v_dense(V); ctx.set_w(W0); ctx.set_h(H0)
            if algo == _lib.ALGO_BNMF:
                ctx.set_lambda(0.2, 0.2)
            ferr, _, _ = ctx.factorize(5)
            outs.append((ctx.get_w(), ctx.get_h(), ferr))
            ctx.close()
    for a, b in zip(outs[:3], outs[3:]):
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])


@pytest.mark.parametrize("algo_name,m,n,k", [
    ("NMF", 4096, 128, 16), ("NMF", 4160, 256, 30), ("NMF", 8192, 1024, 64), ("NMF", 2048, 384, 128),
    ("BNMF", 4096, 256, 64), ("RNMF", 4100, 128, 100), ("NMFALS", 4096, 256, 48), ("NMF", 2048, 256, 200),
])
def test_rowgemm_stream_and_rowgemm_are_bit_identical(pm, algo_name, m, n, k):
    """`k_rowgemm_stream` (A straight into registers, requests interleaved with the MFMAs; contractions that are a
    multiple of 128 wide) forms every accumulator in the same order as `k_rowgemm`, and `k_colgemm_stream` (the
    mirror for W^T V | W^T W: 32 < num_bases <= 64 and blocks of 128 bases) in the same order as `k_colgemm`: the W step (nmf.py:128-132,
    bnmf.py:87-90, rnmf.py:109-115; the plain product of nmfals.py:88 and of base blocks beyond 128) must come out
    bit for bit the same whichever of the two runs, ragged row counts and partly filled base tiles included."""
    from pymf_amd import _lib
    rs = np.random.RandomState(m + n + k)
    V = rs.random_sample((m, n)).astype(np.float32)
    if algo_name == "BNMF":
        V = (V < 0.3).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = rs.random_sample((k, n)).astype(np.float32)
    outs = []
    for stream in (1, 0):
        c = _lib.Context(getattr(_lib, "ALGO_" + algo_name), m, n, k)
        c.set_option("force_tiled", 1)
        c.set_option("rowgemm_stream", stream)
        c.set_option("colgemm_stream", stream)       # k_colgemm_stream vs k_colgemm: the same, for the H step's partials
        if algo_name == "BNMF":
            c.set_lambda(0.3, 0.2)
        if algo_name == "RNMF":
            c.set_lambda(0.7, 0.7)
        c.set_v_dense(V); c.set_w(W0); c.set_h(H0)
        if algo_name == "RNMF":
            c.rnmf_update_s()                          # rnmf.py:96-98: the W step reads S - data
        for _ in range(2):
            c.update_w()
            c.update_h()
        outs.append((c.get_w(), c.get_h()))
    assert np.isfinite(outs[0][0]).all() and np.isfinite(outs[0][1]).all()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_fused_vs_tiled_random_shapes(pm):
    """Randomised sweep of (m, n, k) over the fused-kernel shape space, including row counts where
    waves get 0, 1 or uneven numbers of 16-row blocks: fused factorize == tiled hooks."""
    from pymf_amd import _lib
    rs = np.random.RandomState(2024)
    shapes = [(16, 64, 16), (15, 33, 7), (4096 * 16 + 16, 256, 64), (16400, 192, 48), (1023 * 16, 128, 64),
              (1024 * 16, 64, 32), (1025 * 16 + 3, 256, 17), (70000, 250, 60), (33, 256, 64),
              (5000, 300, 16), (5000, 384, 10), (9000, 320, 32), (3000, 190, 64), (20000, 130, 20),
              (5000, 512, 32), (3000, 400, 20), (7000, 500, 9), (4099, 384, 31), (17, 450, 32), (65536 + 48, 512, 17)]
    for _ in range(10):
        shapes.append((int(rs.randint(1, 40000)), int(rs.randint(1, 513)), int(rs.randint(1, 65))))
    for (m, n, k) in shapes:
        V = rs.random_sample((m, n)).astype(np.float32)
        W0 = rs.random_sample((m, k)).astype(np.float32)
        H0 = rs.random_sample((k, n)).astype(np.float32)
        for algo in (_lib.ALGO_NMF, _lib.ALGO_SNMF):
            if algo == _lib.ALGO_SNMF and 2 * k > n:
                continue              # H H^T (k x k, rank <= n) is singular or nearly so: inv() is noise
            a = _lib.Context(algo, m, n, k)
            if not a.path_name.startswith("k_nmf_fused"):
                a.close()
                continue
            a.set_v_dense(V); a.set_w(W0); a.set_h(H0)
            fa, _, _ = a.factorize(2, compute_err=True)
            b = _lib.Context(algo, m, n, k)
            b.set_option("force_tiled", 1)     # the any-shape kernels on the one-pass kernels' shapes
            b.set_v_dense(V); b.set_w(W0); b.set_h(H0)
            fb = []
            for _ in range(2):
                b.update_w(); b.update_h()
                b.set_w(b.get_w())             # stale (P | S): direct residual pass
                fb.append(b.frobenius())
            tol = 2e-5 if algo == _lib.ALGO_NMF else 5e-4     # SNMF: inv(H H^T) amplifies for k ~ n
            assert rel_fro(a.get_h(), b.get_h(), what="a.get_h()") < tol, (algo, m, n, k)
            assert rel_fro(a.get_w(), b.get_w(), what="a.get_w()") < tol, (algo, m, n, k)
            if fb[-1] > 1e-3 * np.linalg.norm(V):
                close(fa, np.array(fb), rtol=1e-4, err_msg=str((algo, m, n, k)), what="fa")
            a.close(); b.close()


def test_fifty_iterations_drift(pm):
    """SURVEY 8(d) tolerance statement: after 50 iterations against the float64-default reference
    path, ||X - X_ref||_F / ||X_ref||_F <= 2e-5 and |ferr - ferr_ref| / ferr_ref <= 1e-5 (8192 x 256, k = 64)."""
    from oracle import NMFOracle
    rs = np.random.RandomState(1234)
    V = rs.random_sample((8192, 256)).astype(np.float32)
    np.random.seed(42)
    W0, H0 = np.random.random((8192, 64)), np.random.random((64, 256))
    mdl = _run(pm.NMF, V, 64, 50, W0, H0)
    ref = NMFOracle(V, num_bases=64)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=50)
    assert len(mdl.ferr) == len(ref.ferr) == 50
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 4e-6
    assert rel_fro(mdl.H, ref.H, what="mdl.H") < 2e-6
    close(mdl.ferr, ref.ferr, rtol=2e-8, what="mdl.ferr")


def test_fixed_basis_loop_reuses_partials(pm):
    """factorize(compute_w=False): the reference's 'coefficients for an existing basis' use
    (nmf.py:56-65).  W never changes, so (W^T V | W^T W) is formed once and every further iteration
    is a k x n sized kernel -- results must equal the oracle, which recomputes everything."""
    from oracle import NMFOracle, SNMFOracle
    rs = np.random.RandomState(6)
    V = rs.random_sample((5000, 200)).astype(np.float32)
    W0, H0 = rs.random_sample((5000, 40)), rs.random_sample((40, 200))
    for cls, ocls in ((pm.NMF, NMFOracle), (pm.SNMF, SNMFOracle)):
        mdl = cls(V, num_bases=40)
        mdl.W, mdl.H = W0.copy(), H0.copy()
        mdl.factorize(niter=25, compute_w=False)
        ref = ocls(V, num_bases=40)
        ref.W, ref.H = W0.copy(), H0.copy()
        ref.factorize(niter=25, compute_w=False)
        assert rel_fro(mdl.H, ref.H, what="mdl.H") < 2e-6
        close(mdl.ferr, ref.ferr, rtol=2e-8, what="mdl.ferr")
        np.testing.assert_array_equal(mdl.W, W0)                 # W untouched
        mdl.W = W0 * 1.5                                         # a new basis must invalidate the cache
        ref.W = W0 * 1.5
        mdl.factorize(niter=3, compute_w=False)
        ref.factorize(niter=3, compute_w=False)
        assert rel_fro(mdl.H, ref.H, what="mdl.H") < 2e-6


def test_fixed_coefficients_loop_reuses_numerator(pm):
    """factorize(compute_h=False) (tests/test_pymf.py:92): H never changes, so V H^T is formed in the
    first iteration and read back afterwards -- results must equal the oracle."""
    from oracle import NMFOracle
    rs = np.random.RandomState(9)
    V = rs.random_sample((3000, 300)).astype(np.float32)
    W0, H0 = rs.random_sample((3000, 24)), rs.random_sample((24, 300))
    mdl = pm.NMF(V, num_bases=24)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    mdl.factorize(niter=12, compute_h=False)
    ref = NMFOracle(V, num_bases=24)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=12, compute_h=False)
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 9e-6
    close(mdl.ferr, ref.ferr, rtol=2e-9, what="mdl.ferr")
    np.testing.assert_array_equal(mdl.H, H0)
    mdl.H = H0 * 0.7                                         # new coefficients invalidate the cache
    ref.H = H0 * 0.7
    mdl.factorize(niter=4, compute_h=False)
    ref.factorize(niter=4, compute_h=False)
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 9e-6
    mdl.factorize(niter=3)                                   # and the full loop still works afterwards
    ref.factorize(niter=3)
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 8e-6 and rel_fro(mdl.H, ref.H, what="mdl.H") < 5e-7


def test_single_element_edit_is_noticed(pm):
    from oracle import NMFOracle
    rs = np.random.RandomState(13)
    V = rs.random_sample((20000, 64)).astype(np.float32)
    mdl = pm.NMF(V, num_bases=8)
    np.random.seed(3)
    mdl.factorize(niter=2)
    ref = NMFOracle(V, num_bases=8)
    ref.W, ref.H = mdl.W.copy(), mdl.H.copy()
    mdl.W[12345, 3] = 0.0                                    # one poke in a 160 000-element array
    ref.W[12345, 3] = 0.0
    mdl.factorize(niter=2)
    ref.factorize(niter=2)
    assert mdl.W[12345, 3] == 0.0                            # multiplicative updates keep a zero at zero
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 6e-7 and rel_fro(mdl.H, ref.H, what="mdl.H") < 3e-7


@pytest.mark.parametrize("algo_name", ["NMF", "SNMF", "BNMF"])
def test_free_running_loop_matches_stepwise_loop(pm, algo_name):
    """pmf_factorize enqueues chunks of iterations with the error and the convergence test of
    nmf.py:134-139 on the device; the outcome must be the one of the iteration-by-iteration loop
    (hooks + host-side test), including WHERE a convergence stops it (and, for BNMF, the penalty
    weights the reference's schedule has reached by then)."""
    from pymf_amd import _lib
    algo = getattr(_lib, "ALGO_" + algo_name)
    rs = np.random.RandomState(77)
    m, n, k = 4096, 128, 16
    V = rs.random_sample((m, n)).astype(np.float32)
    if algo_name == "BNMF":
        V = (V < 0.3).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = rs.random_sample((k, n)).astype(np.float32)
    for niter, eps in ((37, 1e-8), (60, 2e-5), (200, 1e-6)):
        a = _lib.Context(algo, m, n, k)
        a.set_v_dense(V); a.set_w(W0); a.set_h(H0)
        if algo_name == "BNMF":
            a.set_lambda(1.0 / niter, 1.0 / niter)
        if algo_name == "SNMF":
            a.set_option("snmf_gram", 1)
        fa, done_a, conv_a = a.factorize(niter, conv_eps=eps)
        b = _lib.Context(algo, m, n, k)
        b.set_v_dense(V); b.set_w(W0); b.set_h(H0)
        if algo_name == "BNMF":
            b.set_lambda(1.0 / niter, 1.0 / niter)
        if algo_name == "SNMF":          # one-iteration calls would pick the pass-per-iteration form: same form on both sides
            b.set_option("snmf_gram", 1)
        fb, done_b, conv_b = [], 0, -1
        for i in range(niter):                           # the reference loop, one C call per iteration
            f1, _, _ = b.factorize(1, conv_eps=0.0)
            fb.append(f1[0]); done_b += 1
            if i > 1 and abs(fb[i] - fb[i - 1]) / n < eps:
                conv_b = i
                break
        assert (done_a, conv_a) == (done_b, conv_b), (niter, eps)
        close(fa[:done_a], fb, rtol=1e-12, what="fa[:done_a]")
        np.testing.assert_array_equal(a.get_w(), b.get_w())
        np.testing.assert_array_equal(a.get_h(), b.get_h())
        if algo_name == "BNMF":
            assert a.get_lambda() == b.get_lambda()
        # wherever the loop stopped, the context must be in a state the single hooks can go on from
        # (inside the loop G = H H^T lives as per-workgroup partial sums)
        a.update_w(); b.update_w()
        a.update_h(); b.update_h()
        np.testing.assert_array_equal(a.get_w(), b.get_w())
        np.testing.assert_array_equal(a.get_h(), b.get_h())
        f2a, _, _ = a.factorize(3, conv_eps=0.0)
        f2b, _, _ = b.factorize(3, conv_eps=0.0)
        close(f2a, f2b, rtol=1e-12, what="f2a")
        a.close(); b.close()


@pytest.mark.parametrize("shape,k", [((20000, 256), 64), ((9000, 512), 32), ((7000, 192), 48), ((5000, 700), 16)])
def test_results_are_reproducible_bit_for_bit(pm, shape, k):
    """Every reduction runs in a fixed order (per-wave chains, cross-wave sums through LDS, float64 slab
    sums, partial Gram matrices added by workgroup index): two runs give identical bits."""
    from pymf_amd import _lib
    rs = np.random.RandomState(k)
    V = rs.random_sample(shape).astype(np.float32)
    W0 = rs.random_sample((shape[0], k)).astype(np.float32)
    H0 = rs.random_sample((k, shape[1])).astype(np.float32)
    outs = []
    for _ in range(2):
        ctx = _lib.Context(_lib.ALGO_NMF, shape[0], shape[1], k)
        ctx.set_v_dense(V); ctx.set_w(W0); ctx.set_h(H0)
        ferr, _, _ = ctx.factorize(12)
        outs.append((ctx.get_w(), ctx.get_h(), ferr.copy()))
        ctx.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("cls_name,shape,k", [("NMF", (3000, 400), 128), ("NMF", (2000, 700), 40), ("BNMF", (2500, 450), 100),
                                              ("SNMF", (3000, 400), 100), ("NMF", (1500, 1100), 20)])
def test_tiled_kernels_vs_oracle(pm, cls_name, shape, k):
    """Shapes the one-pass kernel does not take (k > 64, or wider than its LDS budget) run on the
    two-pass tiled kernels (k_rowgemm / k_colgemm): same tolerances against the oracle."""
    import oracle
    rs = np.random.RandomState(shape[1] + k)
    V = rs.random_sample(shape).astype(np.float32)
    if cls_name == "BNMF":
        V = (V < 0.3).astype(np.float32)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    mdl = getattr(pm, cls_name)(V, num_bases=k)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    mdl.factorize(niter=4)
    assert mdl._ctx.path_name == "tiled"
    o = getattr(oracle, cls_name + "Oracle")(V, num_bases=k)
    o.W, o.H = W0.copy(), H0.copy()
    o.factorize(niter=4)
    tol = 5e-4 if cls_name == "SNMF" else 5e-5
    assert rel_fro(mdl.W, o.W, what="mdl.W") < tol and rel_fro(mdl.H, o.H, what="mdl.H") < tol
    close(mdl.ferr, o.ferr, rtol=2e-8, what="mdl.ferr")


@pytest.mark.parametrize("shape,k", [((2000, 300), 200), ((1500, 500), 129), ((700, 900), 300), ((3000, 64), 256)])
def test_nmf_more_than_128_bases(pm, shape, k):
    """num_bases > 128 (NMF): blocks of 128 bases on the tiled kernels; error through the trace identity."""
    from oracle import NMFOracle
    rs = np.random.RandomState(k)
    V = rs.random_sample(shape).astype(np.float32)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    mdl = pm.NMF(V, num_bases=k)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    mdl.factorize(niter=4)
    o = NMFOracle(V, num_bases=k)
    o.W, o.H = W0.copy(), H0.copy()
    o.factorize(niter=4)
    assert rel_fro(mdl.W, o.W, what="mdl.W") < 3e-6 and rel_fro(mdl.H, o.H, what="mdl.H") < 2e-6
    close(mdl.ferr, o.ferr, rtol=7e-9, what="mdl.ferr")       # trace identity only at this width
    mdl.update_w(); o.update_w()
    mdl.update_h(); o.update_h()
    assert rel_fro(mdl.W, o.W, what="mdl.W") < 4e-6 and rel_fro(mdl.H, o.H, what="mdl.H") < 2e-6
    assert abs(mdl.frobenius_norm() - o.frobenius_norm()) <= 5e-5 * o.frobenius_norm()


@pytest.mark.parametrize("cls_name", ["NMF", "BNMF", "SNMF"])
def test_error_after_a_w_only_step_is_not_stale(pm, cls_name):
    """factorize() leaves the trace terms of its last H step behind; a later update_w() / W-only loop
    changes W (and, on fused shapes, recomputes W^T V | W^T W): the error must come from the NEW state."""
    import oracle
    rs = np.random.RandomState(2)
    V = rs.random_sample((200, 65)).astype(np.float32)
    if cls_name == "BNMF":
        V = (V < 0.3).astype(np.float32)
    k = 12
    W0, H0 = rs.random_sample((200, k)), rs.random_sample((k, 65))
    a = getattr(pm, cls_name)(V, num_bases=k); a.W, a.H = W0.copy(), H0.copy()
    o = getattr(oracle, cls_name + "Oracle")(V, num_bases=k); o.W, o.H = W0.copy(), H0.copy()
    a.factorize(niter=5); o.factorize(niter=5)
    a.factorize(niter=7, compute_h=False); o.factorize(niter=7, compute_h=False)
    # SNMF's W step with H fixed is a closed form: the error repeats and the loop stops at i == 2
    assert len(a.ferr) == len(o.ferr) == (2 if cls_name == "SNMF" else 7)
    close(a.ferr, o.ferr, rtol=1e-7, what="a.ferr")     # (float32-stored W and H: 6e-8 each; SURVEY 8(d) states 1e-5)
    tol = 5e-4 if cls_name == "SNMF" else 5e-5
    assert rel_fro(a.W, o.W, what="a.W") < tol
    a.update_w(); o.update_w()
    assert abs(a.frobenius_norm() - o.frobenius_norm()) <= 2e-5 * o.frobenius_norm()


@pytest.mark.parametrize("algo_name", ["NMF", "BNMF"])
def test_free_running_fixed_basis_loop_matches_stepwise_loop(pm, algo_name):
    """factorize(compute_w=False) -- coefficients for an existing basis (nmf.py:56-65) -- free-runs as well
    (one H-step kernel per iteration): same outcome as one C call per iteration."""
    from pymf_amd import _lib
    algo = getattr(_lib, "ALGO_" + algo_name)
    rs = np.random.RandomState(5)
    m, n, k = 3000, 192, 24
    V = rs.random_sample((m, n)).astype(np.float32)
    if algo_name == "BNMF":
        V = (V < 0.3).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = rs.random_sample((k, n)).astype(np.float32)
    for niter, eps in ((30, 1e-8), (80, 1e-4), (300, 1e-6)):
        outs = []
        for stepwise in (False, True):
            c = _lib.Context(algo, m, n, k)
            c.set_v_dense(V); c.set_w(W0); c.set_h(H0)
            if algo_name == "BNMF":
                c.set_lambda(1.0 / niter, 1.0 / niter)
            if not stepwise:
                f, done, conv = c.factorize(niter, compute_w=False, conv_eps=eps)
                f = list(f[:done])
            else:
                f, done, conv = [], 0, -1
                for i in range(niter):
                    f1, _, _ = c.factorize(1, compute_w=False, conv_eps=0.0)
                    f.append(f1[0]); done += 1
                    if i > 1 and abs(f[i] - f[i - 1]) / n < eps:
                        conv = i
                        break
            outs.append((done, conv, np.array(f), c.get_h(), c.get_w(), c.get_lambda() if algo_name == "BNMF" else None))
            c.close()
        assert outs[0][:2] == outs[1][:2], (niter, eps, outs[0][:2], outs[1][:2])
        close(outs[0][2], outs[1][2], rtol=1e-12, what="outs[0][2]")
        np.testing.assert_array_equal(outs[0][3], outs[1][3])
        np.testing.assert_array_equal(outs[0][4], W0)
        assert outs[0][5] == outs[1][5]


# ---- the plugin API: overridden hooks are called (nmf.py:182-202 is a template method) -----------
def _plugin_classes(base, obase):
    """The same user plug-in written against the product class and against the oracle class."""
    def make(b):
        class Plugin(b):
            def __init__(self, *a, **kw):
                b.__init__(self, *a, **kw)
                self.calls = {"update_w": 0, "update_h": 0, "frobenius_norm": 0, "converged": 0}

            def update_w(self):
                self.calls["update_w"] += 1
                b.update_w(self)
                self.W[:, 0] *= 0.5                       # a user rule on top of the built-in step (in place)

            def update_h(self):
                self.calls["update_h"] += 1
                b.update_h(self)

            def frobenius_norm(self):
                self.calls["frobenius_norm"] += 1
                return b.frobenius_norm(self)

            def converged(self, i):
                self.calls["converged"] += 1
                return i >= 4                            # user stopping rule: stop at the fifth iteration
        return Plugin
    return make(base), make(obase)


@pytest.mark.parametrize("cls_name", ["NMF", "SNMF", "BNMF"])
def test_overridden_hooks_are_called_like_the_reference(pm, cls_name):
    import oracle
    P, O = _plugin_classes(getattr(pm, cls_name), getattr(oracle, cls_name + "Oracle"))
    rs = np.random.RandomState(31)
    V = rs.random_sample((700, 256)).astype(np.float32)
    if cls_name == "BNMF":
        V = (V < 0.3).astype(np.float32)
    W0, H0 = rs.random_sample((700, 24)), rs.random_sample((24, 256))
    a, o = P(V, num_bases=24), O(V, num_bases=24)
    a.W, a.H = W0.copy(), H0.copy()
    o.W, o.H = W0.copy(), H0.copy()
    a.factorize(niter=9)
    o.factorize(niter=9)
    assert a.calls == o.calls == {"update_w": 5, "update_h": 5, "frobenius_norm": 5, "converged": 3}
    assert len(a.ferr) == len(o.ferr) == 4                   # the user's converged() fired at i == 4
    tol = 5e-5 if cls_name == "SNMF" else TOL_X
    assert rel_fro(a.W, o.W, what="a.W") < tol and rel_fro(a.H, o.H, what="a.H") < tol
    close(a.ferr, o.ferr, rtol=2e-8, what="a.ferr")


def test_hook_loop_equals_one_call_loop(pm):
    """A subclass that overrides a hook WITHOUT changing it must get the bits of the one-call path
    (the hook loop runs the same kernels; W and H stay on the device between the hooks)."""
    class Same(pm.NMF):
        n_w = 0

        def update_w(self):
            Same.n_w += 1
            pm.NMF.update_w(self)

    rs = np.random.RandomState(5)
    V = rs.random_sample((9000, 256)).astype(np.float32)
    W0, H0 = rs.random_sample((9000, 64)), rs.random_sample((64, 256))
    a, b = pm.NMF(V, num_bases=64), Same(V, num_bases=64)
    for mdl in (a, b):
        mdl.W, mdl.H = W0.copy(), H0.copy()
        w_obj = mdl.W
        mdl.factorize(niter=7)
        assert mdl.W is w_obj                                    # still the user's array, updated in place
    assert Same.n_w == 7
    np.testing.assert_array_equal(a.W, b.W)
    np.testing.assert_array_equal(a.H, b.H)
    # the error: same (P | S), H and ||V||^2, but the trace terms are summed by k_trace_terms in the hook
    # loop and inside k_nmf_h_gram in the one-call loop -- float64 sums in a different order
    close(a.ferr, b.ferr, rtol=3e-9, what="a.ferr")


def test_instance_level_hook_and_show_progress(pm, caplog):
    """A hook replaced on the INSTANCE is honoured too; show_progress=True logs every iteration as it
    runs (nmf.py:191-194) -- the reference's message format."""
    import logging
    from oracle import NMFOracle
    rs = np.random.RandomState(6)
    V = rs.random_sample((300, 64)).astype(np.float32)
    W0, H0 = rs.random_sample((300, 8)), rs.random_sample((8, 64))
    mdl = pm.NMF(V, num_bases=8)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    seen = []
    mdl.converged = lambda i: seen.append(i) or False
    with caplog.at_level(logging.INFO, logger="pymf"):
        mdl.factorize(niter=5, show_progress=True)
    assert seen == [2, 3, 4]
    msgs = [r.getMessage() for r in caplog.records if r.name == "pymf"]
    assert [m.split(" FN:")[0] for m in msgs] == ["Iteration %d/5" % (i + 1) for i in range(5)]
    assert [float(m.split(" FN:")[1]) for m in msgs] == [float(x) for x in mdl.ferr]
    ref = NMFOracle(V, num_bases=8)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=5)
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 8e-7 and rel_fro(mdl.H, ref.H, what="mdl.H") < 5e-7
    close(mdl.ferr, ref.ferr, rtol=7e-8, what="mdl.ferr")


# ---- the device copies follow in-place edits of the host arrays -----------------------------------
def test_row_swap_and_sum_preserving_edits_are_noticed(pm):
    from oracle import NMFOracle
    rs = np.random.RandomState(14)
    V = rs.random_sample((2000, 128)).astype(np.float32)
    mdl = pm.NMF(V, num_bases=16)
    np.random.seed(4)
    mdl.factorize(niter=2)
    ref = NMFOracle(V, num_bases=16)
    ref.W, ref.H = mdl.W.copy(), mdl.H.copy()
    for m_ in (mdl, ref):
        m_.W[[5, 1500]] = m_.W[[1500, 5]]                    # swap two rows: same sum, same last element
        m_.H[[0, 9]] = m_.H[[9, 0]]                          # permute two bases of H
        m_.W[0, 0] += 0.25
        m_.W[1, 0] -= 0.25                                   # sum-preserving poke
        m_.factorize(niter=3)
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 9e-7 and rel_fro(mdl.H, ref.H, what="mdl.H") < 3e-7
    close(mdl.ferr, ref.ferr, rtol=2e-8, what="mdl.ferr")


@pytest.mark.parametrize("cls_name", ["NMF", "SNMF"])
def test_data_edited_in_place_is_noticed(pm, cls_name):
    """The reference reads self.data[:, :] afresh in every hook (nmf.py:123,129): editing `data` in
    place between two calls must reach the device (V, ||V||^2 and the cached partial sums)."""
    import oracle
    rs = np.random.RandomState(15)
    V = rs.random_sample((1500, 128)).astype(np.float32)
    W0, H0 = rs.random_sample((1500, 16)), rs.random_sample((16, 128))
    Va, Vo = V.copy(), V.copy()
    a = getattr(pm, cls_name)(Va, num_bases=16)
    o = getattr(oracle, cls_name + "Oracle")(Vo, num_bases=16)
    for m_ in (a, o):
        m_.W, m_.H = W0.copy(), H0.copy()
        m_.factorize(niter=3)
    for m_, X in ((a, Va), (o, Vo)):
        X *= 1.7                                             # in place: same object, new contents
        X[100:200, :] = 0.25
    assert abs(a.frobenius_norm() - o.frobenius_norm()) <= 2e-5 * o.frobenius_norm()
    a.factorize(niter=3); o.factorize(niter=3)
    tol = 5e-5 if cls_name == "SNMF" else TOL_X
    assert rel_fro(a.W, o.W, what="a.W") < tol and rel_fro(a.H, o.H, what="a.H") < tol
    close(a.ferr, o.ferr, rtol=4e-8, what="a.ferr")
    # check_data = False: the upload happens once per object until invalidate_data() is called
    a.check_data = False
    Va[:, 3] = 0.0; Vo[:, 3] = 0.0
    a.invalidate_data()
    a.update_h(); o.update_h()
    assert rel_fro(a.H, o.H, what="a.H") < tol
    assert np.all(a.H[:, 3] == 0.0) or cls_name == "SNMF"


@pytest.mark.parametrize("cls_name", ["NMF", "SNMF", "NMFALS"])
def test_digest_of_data_beside_the_loop_equals_checking_first(pm, cls_name, monkeypatch):
    """Round 5: the digest of a large `data` runs on a second thread WHILE pmf_factorize iterates on the
    resident copy; unchanged bytes: nothing else happens; edited bytes: the loop is stopped (pmf_abort), W / H are put back
    from the copies kept on the device, the new bytes go up and the call starts again.  Both must give the bits of the
    check-first path (nmf.py:123,129: the reference reads self.data[:,:] afresh in every hook)."""
    from pymf_amd import nmf as nmfmod
    rs = np.random.RandomState(21)
    V = (rs.random_sample((4096, 128)) - (0.3 if cls_name == "SNMF" else 0.0)).astype(np.float32)
    made, restarted = [], []
    orig_init, orig_restart = nmfmod._LateDataCheck.__init__, nmfmod._LateDataCheck.restart
    monkeypatch.setattr(nmfmod._LateDataCheck, "__init__", lambda self, *a: (made.append(1), orig_init(self, *a))[1])
    monkeypatch.setattr(nmfmod._LateDataCheck, "restart", lambda self: (restarted.append(1), orig_restart(self))[1])

    def run(late):
        np.random.seed(5)
        m_ = getattr(pm, cls_name)(V.copy(), num_bases=8)
        m_._LATE_DATA_CHECK, m_._LATE_DATA_CHECK_MIN_BYTES = late, 0
        m_.factorize(niter=3)                                  # first call: the upload
        m_.factorize(niter=4)                                  # data unchanged
        a = (m_.W.copy(), m_.H.copy(), m_.ferr.copy())
        m_.data[7, 3] += 0.25
        m_.data[100:200] *= 0.5                                # edited in place: same object, new bytes
        m_.factorize(niter=5)
        b = (m_.W.copy(), m_.H.copy(), m_.ferr.copy())
        m_.factorize(niter=2, compute_err=False)               # and on from there, unchanged again
        return a, b, (m_.W.copy(), m_.H.copy())
    first = run(False)
    assert not made
    beside = run(True)
    assert len(made) == 3 and len(restarted) == 1, (made, restarted)
    for x, y in zip(first, beside):
        for u, v in zip(x, y):
            assert u.shape == v.shape and np.array_equal(u, v)


@pytest.mark.parametrize("shape,k,sparse", [((6000, 256), 64, False), ((3000, 320), 20, False), ((5000, 128), 128, True),
                                            ((4000, 200), 48, True), ((2500, 700), 33, False)])
def test_snmf_gram_space_loop_equals_pass_per_iteration(pm, shape, k, sparse):
    """SNMF factorize() in Gram space (P = M^T V^T V, S = P M; W once at the end) against the form that
    passes over V in every iteration (option snmf_gram = 0) and against the oracle: same W, H, ferr."""
    import scipy.sparse as sp
    from pymf_amd import _lib
    from oracle import SNMFOracle
    rs = np.random.RandomState(shape[1] + k)
    if sparse:
        Vs = sp.random(shape[0], shape[1], density=0.02, format="csr", dtype=np.float32, random_state=rs)
        Vd = np.asarray(Vs.toarray(), dtype=np.float32)
    else:
        Vd = (rs.random_sample(shape) - 0.3).astype(np.float32)
    W0 = rs.random_sample((shape[0], k)).astype(np.float32)
    H0 = (rs.random_sample((k, shape[1])) + 0.1).astype(np.float32)
    outs = []
    for gram in (1, 0, 2):             # Gram space / one pass over V per iteration / Gram space + W written every iteration
        c = _lib.Context(_lib.ALGO_SNMF, shape[0], shape[1], k)
        if sparse:
            c.set_v_csr(Vs.indptr, Vs.indices, Vs.data)
        else:
            c.set_v_dense(Vd)
        c.set_w(W0); c.set_h(H0)
        c.set_option("snmf_gram", gram)
        ferr, done, conv = c.factorize(5, compute_err=not sparse)
        assert done == 5
        outs.append((c.get_w(), c.get_h(), ferr))
        # the context stays usable hook by hook after either loop
        c.update_w(); c.update_h()
        outs[-1] += (c.get_w(), c.get_h())
        c.close()
    o = SNMFOracle(Vd, num_bases=k); o.W, o.H = W0.astype(np.float64), H0.astype(np.float64)
    o.factorize(niter=5, compute_err=not sparse)
    # k ~ n: H H^T is near-singular (cond ~ 1e7) and W = V pinv(H) amplifies any error of H by sigma_max / sigma_min ~ 3e3.  The
    # Gram-space loop forms P and S in float64 and (round 6) keeps H in float64 on the device: it meets the stated tolerances
    # (W 2e-5 / 5e-5 here with the float32 product W = V M, H 5e-6 ... 2e-5) whatever the conditioning; the
    # pass-per-iteration form takes P = W^T V from the float32 MFMA (1e-6 relative), which the next W step amplifies
    ill = 2 * k > shape[1]
    tol = 5e-3 if ill else 2e-5
    assert rel_fro(outs[0][0], o.W, what="gram W vs oracle") < 5e-5
    assert rel_fro(outs[0][1], o.H, what="gram H vs oracle") < 2e-5
    assert rel_fro(outs[0][0], outs[1][0], what="gram W vs pass-per-iteration W") < (tol if ill else 5e-5)
    assert rel_fro(outs[0][1], outs[1][1], what="gram H vs pass-per-iteration H") < tol
    assert rel_fro(outs[0][3], outs[1][3], what="W after the hooks that follow") < (tol if ill else 5e-5)
    assert rel_fro(outs[0][4], outs[1][4], what="H after the hooks that follow") < tol
    np.testing.assert_array_equal(outs[0][0], outs[2][0])        # option 2 only writes W more often: same bits
    np.testing.assert_array_equal(outs[0][1], outs[2][1])
    if not sparse:
        close(outs[0][2], o.ferr, rtol=2e-8, what="gram ferr vs oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("shape,k,density", [((5000, 128), 128, 0.02), ((3000, 96), 40, 0.05), ((2000, 320), 64, 0.01)])
def test_csr_snmf_is_reproducible_run_to_run(pm, shape, k, density):
    """C = V^T V of CSR data is accumulated by many waves at once (pmf_csr.h: k_csr_gram).  As float64 atomics the order of the
    additions was the library's one run-to-run freedom (1e-16 relative); with SNMF's H in float64 and cond(H H^T) ~ 1e7 that
    reached the last bits of W (first seen as a 3 % mismatch between the two Gram-space loop forms).  Round 6 accumulates the
    exact float32 x float32 products in two-limb FIXED POINT with integer atomics: any order, same bits.  Four contexts, same
    inputs: W, the float32 H and the float64 H must be identical; values spanning 12 orders of magnitude and both signs; and C
    itself (through the factors) must match the float64 oracle as closely as before."""
    import scipy.sparse as sp
    from pymf_amd import _lib
    from oracle import SNMFOracle
    rs = np.random.RandomState(shape[1] + 7 * k)
    Vs = sp.random(shape[0], shape[1], density=density, format="csr", dtype=np.float32, random_state=rs)
    Vs.data = (Vs.data - 0.3).astype(np.float32)
    Vs.data[::7] *= np.float32(1e-6)                      # tiny entries beside ordinary ones: their products sit far down the limbs
    Vs.data[::11] *= np.float32(1e3)
    W0 = rs.random_sample((shape[0], k))
    H0 = rs.random_sample((k, shape[1])) + 0.1
    outs = []
    for rep in range(4):
        c = _lib.Context(_lib.ALGO_SNMF, shape[0], shape[1], k)
        c.set_v_csr(Vs.indptr, Vs.indices, Vs.data)
        c.set_w(W0); c.set_h(H0)
        c.set_option("snmf_gram", 1 if rep % 2 == 0 else 2)
        _, done, _ = c.factorize(6, compute_err=False)
        assert done == 6
        Hd = np.empty((k, shape[1]))
        assert c.get_h_into(Hd)
        outs.append((c.get_w(), c.get_h(), Hd))
        c.close()
    for o in outs[1:]:
        np.testing.assert_array_equal(o[0], outs[0][0])
        np.testing.assert_array_equal(o[1], outs[0][1])
        np.testing.assert_array_equal(o[2], outs[0][2])
    ref = SNMFOracle(np.asarray(Vs.toarray(), dtype=np.float32), num_bases=k)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=6, compute_err=False)
    assert rel_fro(outs[0][2], ref.H, what="float64 H vs oracle") < 2e-6
    assert rel_fro(outs[0][0], ref.W, what="W vs oracle") < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3, 15, 16, 17, 31, 33, 47, 63, 64, 65, 97, 111, 127, 128])
def test_snmf_w_step_at_the_edges_of_the_inverse(pm, k):
    """W = V H^T inv(H H^T) (snmf.py:67-70) hook by hook at base counts around the tiles of k_inverse_spd_mfma: the in-wave
    16 x 16 inverse eliminates two pivots per step (rank-2 updates, closed-form 2 x 2 blocks), so odd k puts an identity row of
    the padding into the last pair, k = 16 j +- 1 moves that pair across a tile boundary, and two nearly parallel rows of H
    make one 2 x 2 pivot block itself ill-conditioned (det = alpha gamma - beta^2 cancels)."""
    from pymf_amd import _lib
    from oracle import SNMFOracle
    rs = np.random.RandomState(4000 + k)
    m, n = 600, max(2 * k + 8, 48)
    V = (rs.random_sample((m, n)) - 0.4).astype(np.float32)
    H0 = (rs.random_sample((k, n)) + 0.05).astype(np.float32)
    if k >= 2:                                             # rows 0 and 1 nearly parallel: the FIRST pivot pair is the hard one
        H0[1] = (H0[0] * 1.25 + 2e-2 * rs.random_sample(n)).astype(np.float32)
    if k >= 18:                                            # ... and one pair in a later tile, straddling an even / odd boundary
        H0[17] = (H0[16] * 0.8 + 2e-2 * rs.random_sample(n)).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    c = _lib.Context(_lib.ALGO_SNMF, m, n, k)
    c.set_v_dense(V); c.set_w(W0); c.set_h(H0)
    c.update_w()
    Wd = c.get_w()
    c.close()
    o = SNMFOracle(V, num_bases=k); o.W, o.H = W0.astype(np.float64), H0.astype(np.float64)
    o.update_w()
    cond = np.linalg.cond(H0.astype(np.float64) @ H0.astype(np.float64).T)
    # float32 storage of M^T = inv(H H^T) H and of W, a float32-MFMA product over n columns: measured 1.1e-7 (k = 1) ... 3.1e-7
    # (k = 128, cond 8e6) -- the float64 inverse does not show at all; a wrong pivot is an O(1) error
    tol = 2e-6
    assert np.all(np.isfinite(Wd))
    assert rel_fro(Wd, o.W, what="W after update_w, k = %d (cond %.1e)" % (k, cond)) < tol


# ---- BASELINE's other configs at FULL size: size-independent properties (cfg4's are above) ----------
def test_pipelined_w_write_of_the_csr_gram_loop_is_bit_identical(pm):
    """snmf_gram = 2 on CSR data writes W = V M in every iteration (snmf.py:67-70 does); since round 4 that write runs on
    a stream of its own beside the k x n sized kernels of the NEXT iteration (option snmf_w_pipe: workgroup slots the
    write leaves free; 0 = stream order).  Same kernels, same operands (M double buffered): W and H must not change by
    a bit, whatever the iteration count's parity, and the context must stay usable hook by hook afterwards."""
    import scipy.sparse as sp
    from pymf_amd import _lib
    m, n, k = 70000, 128, 128
    rs = np.random.RandomState(5)
    Vs = sp.random(m, n, density=0.01, format="csr", dtype=np.float32, random_state=rs)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = (rs.random_sample((k, n)) + 0.1).astype(np.float32)
    outs = {}
    for pipe in (0, 16, 3):
        for niter in (1, 4):
            c = _lib.Context(_lib.ALGO_SNMF, m, n, k)
            c.set_v_csr(Vs.indptr, Vs.indices, Vs.data)
            c.set_w(W0); c.set_h(H0)
            c.set_option("snmf_gram", 2)
            c.set_option("snmf_w_pipe", pipe)
            _, done, _ = c.factorize(niter, compute_err=False)
            assert done == niter
            _, done, _ = c.factorize(1, compute_err=False)          # a second call: buffers and events carry over
            r = [c.get_w(), c.get_h()]
            c.update_w(); c.update_h()
            r += [c.get_w(), c.get_h()]
            outs[(pipe, niter)] = r
            c.close()
    for niter in (1, 4):
        for pipe in (16, 3):
            for a, b in zip(outs[(0, niter)], outs[(pipe, niter)]):
                np.testing.assert_array_equal(a, b)
    assert np.isfinite(outs[(16, 4)][0]).all()


def _synthetic_rows(seed, rows, ncols):
    """Host replica of the library's counter-based U[0,1) fill (u01_from in pmf_dev.h: splitmix64 finaliser
    of seed + golden * (row * ncols + col + 1)) for the given global rows."""
    with np.errstate(over="ignore"):
        idx = (np.asarray(rows, dtype=np.uint64)[:, None] * np.uint64(ncols) + np.arange(ncols, dtype=np.uint64)[None, :])
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0))


@pytest.mark.parametrize("cfg", ["cfg4", "cfg2"])
def test_full_size_update_w_rows_and_h_vs_the_oracle(pm, cfg):
    """the BASELINE shapes AT FULL SIZE against the oracle itself (not HIP against HIP).  After ONE
    update_w on the full matrix every 4097th row of W is compared with oracle.nmf_update_w (nmf.py:128-132, float64) on
    exactly those rows -- the W rule is row-local, and the counter-based V can be rebuilt row by row on the host.  The
    update_h that follows needs all rows: its oracle value (nmf.py:122-126) is formed in float64 from the device's new
    W and the rebuilt V in row chunks -- every row of the matrix takes part."""
    import oracle
    from pymf_amd import _lib
    m, n, k = (1048576, 256, 64) if cfg == "cfg4" else (65536, 512, 32)
    a = _lib.Context(_lib.ALGO_NMF, m, n, k)
    a.fill_v_uniform(1234); a.fill_w_uniform(42); a.fill_h_uniform(43)
    rows = np.arange(0, m, 4097)
    H0 = a.get_h().astype(np.float64)
    W0s = a.get_w()[rows].astype(np.float64)
    Vs = _synthetic_rows(1234, rows, n)
    a.update_w()
    W1 = a.get_w()
    Wref = W0s.copy()
    oracle.nmf_update_w(Vs, Wref, H0.copy())
    assert rel_fro(W1[rows], Wref, what="%s full size: sampled rows of W after one update_w vs oracle.nmf_update_w" % cfg) < 2e-6
    a.update_h()
    H1 = a.get_h()
    # oracle H step from the device's W (float64 accumulation over ALL rows, V rebuilt chunk by chunk)
    P = np.zeros((k, n))
    W1d = W1.astype(np.float64)
    for r0 in range(0, m, 65536):
        r1 = min(m, r0 + 65536)
        P += W1d[r0:r1].T.dot(_synthetic_rows(1234, np.arange(r0, r1), n).astype(np.float64))
    S = W1d.T.dot(W1d)
    Href = H0 * P / (S.dot(H0) + 1e-9)                       # nmf.py:122-126
    assert rel_fro(H1, Href, what="%s full size: H after update_h vs the float64 rule over all rows" % cfg) < 2e-6
    a.close()


def test_full_size_properties_cfg2(pm):
    """cfg2 (NMF 65,536 x 512, k = 32, the two-waves-per-block fused kernel): monotone objective, fused
    one-pass == two-pass tiled hooks, trace-identity error == direct residual, non-negativity."""
    from pymf_amd import _lib
    m, n, k = 65536, 512, 32
    a = _lib.Context(_lib.ALGO_NMF, m, n, k)
    a.fill_v_uniform(1234); a.fill_w_uniform(42); a.fill_h_uniform(43)
    assert a.path_name == "k_nmf_fused<2,4,SPLIT 2>"
    ferr, done, conv = a.factorize(8, compute_err=True)
    assert done == 8 and conv < 0
    assert np.all(np.diff(ferr) <= 1e-6 * ferr[0]), ferr
    b = _lib.Context(_lib.ALGO_NMF, m, n, k)
    b.set_option("force_tiled", 1)               # k_rowgemm<EPI_NMF_W> + k_colgemm
    b.fill_v_uniform(1234); b.fill_w_uniform(42); b.fill_h_uniform(43)
    fb = []
    for _ in range(8):
        b.update_w(); b.update_h()
        b.set_w(b.get_w())                       # forces the direct residual pass
        fb.append(b.frobenius())
    close(ferr, np.array(fb), rtol=5e-8, what="cfg2 ferr: fused + trace identity vs tiled + direct residual")
    assert rel_fro(a.get_h(), b.get_h(), what="cfg2 H fused vs forced-tiled") < 2e-6
    Wa, Wb = a.get_w(), b.get_w()
    assert rel_fro(Wa, Wb, what="cfg2 W fused vs forced-tiled") < 1e-5
    assert float(Wa.min()) >= 0.0 and np.isfinite(Wa).all() and float(a.get_h().min()) >= 0.0
    a.close(); b.close()


def test_full_size_properties_cfg3(pm):
    """cfg3 (NMFALS 262,144 x 1024, k = 64): the 262,144 row QPs of one W half step satisfy their KKT
    conditions (x >= 0, gradient >= -tol, complementarity; every 257th row checked in float64 on the host),
    the objective does not increase over the two half steps, and the column QPs are at a fixed point."""
    from pymf_amd import _lib
    m, n, k = 262144, 1024, 64
    c = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
    c.fill_v_uniform(1234); c.fill_w_uniform(42); c.fill_h_uniform(43)
    f0 = c.frobenius()
    H0 = c.get_h().astype(np.float64)
    c.update_w()
    W1 = c.get_w()
    assert W1.shape == (m, k) and float(W1.min()) >= 0.0 and np.isfinite(W1).all()
    # KKT of the row QPs (nmfals.py:85-97: HA = H H^T, f = H v^T) on a row sample, in float64 on the host;
    # the synthetic V is a counter-based stream, so any row can be rebuilt here
    rows = np.arange(0, m, 257)
    Vs = _synthetic_rows(1234, rows, n).astype(np.float64)
    HA, Fm = H0.dot(H0.T), Vs.dot(H0.T)
    Ws = W1[rows].astype(np.float64)
    g = Ws.dot(HA) - Fm                               # gradient of 1/2 x'HA x - f'x at the solution
    scale = np.abs(Fm).max()
    assert g.min() > -2e-4 * scale, g.min() / scale   # dual feasibility (float32 right-hand sides and storage)
    assert np.abs(Ws * g).max() < 2e-4 * scale * max(1.0, Ws.max())     # complementarity
    # ... and the oracle's minimiser itself on those rows (the KKT band alone would pass with W rows 1e-4 away
    # from it): the exact float64 active-set solve of the same QP, HA = H0 H0^T, FA = -H0 v^T (nmfals.py:88-93)
    import oracle
    Wo = np.array([oracle.nnqp_solve(HA, -Fm[q]) for q in range(len(rows))])
    # (tolerance: DESIGN 4 -- 1e-4 for NMFALS: the right-hand sides V H0^T are float32 MFMA sums over 1 024 columns and the minimiser
    #  amplifies their rounding by the condition of H0 H0^T; 4e-5 on these device-filled inputs, 2.6e-6 on bench.py's seeded ones)
    assert rel_fro(Ws, Wo, what="cfg3 W rows (every 257th) vs oracle.nnqp_solve, full size") < 1e-4
    nz = int(np.sum((Ws == 0) != (Wo == 0)))
    assert nz <= len(rows) * k // 1000, "active sets differ on %d of %d entries (more than 0.1 %%)" % (nz, len(rows) * k)
    f1 = c.frobenius()
    c.update_h()
    f2 = c.frobenius()
    assert f1 <= f0 * (1 + 1e-6) and f2 <= f1 * (1 + 1e-6), (f0, f1, f2)
    H1 = c.get_h().astype(np.float64)
    assert float(H1.min()) >= 0.0
    # fixed-point property of the exact column QPs: a second update_h with unchanged W must not move H
    c.update_h()
    H2 = c.get_h().astype(np.float64)
    assert rel_fro(H2, H1, what="cfg3 H: update_h twice (fixed point of the exact QP)") < 1e-9
    c.close()


def test_full_size_properties_nmfals_128_bases(pm):
    """NMFALS 262,144 x 1024 at k = 128 (k_nnqp_wave, pmf_nnls_wave.h: one wave per QP): the KKT conditions of the row QPs
    of the first W half step from the random start (systems of 50-60 unknowns, 3-5 block-pivoting passes) and of a later,
    warm one, on a row sample in float64; monotone objective; the column QPs at a fixed point."""
    from pymf_amd import _lib
    m, n, k = 262144, 1024, 128
    c = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
    c.fill_v_uniform(1234); c.fill_w_uniform(42); c.fill_h_uniform(43)
    rows = np.arange(0, m, 509)
    Vs = _synthetic_rows(1234, rows, n).astype(np.float64)
    f_prev = c.frobenius()
    for it in range(3):
        H0 = c.get_h().astype(np.float64)
        c.update_w()
        W1 = c.get_w()
        assert float(W1.min()) >= 0.0 and np.isfinite(W1).all()
        HA, Fm = H0.dot(H0.T), Vs.dot(H0.T)
        Ws = W1[rows].astype(np.float64)
        g = Ws.dot(HA) - Fm
        scale = np.abs(Fm).max()
        assert g.min() > -2e-4 * scale, (it, g.min() / scale)
        assert np.abs(Ws * g).max() < 2e-4 * scale * max(1.0, Ws.max()), it
        f1 = c.frobenius()
        c.update_h()
        f2 = c.frobenius()
        assert f1 <= f_prev * (1 + 1e-6) and f2 <= f1 * (1 + 1e-6), (it, f_prev, f1, f2)
        f_prev = f2
    H1 = c.get_h().astype(np.float64)
    assert float(H1.min()) >= 0.0
    c.update_h()
    assert rel_fro(c.get_h().astype(np.float64), H1, what="k = 128 H: update_h twice (fixed point of the exact QP)") < 1e-9
    c.close()


def test_full_size_properties_cfg5(pm):
    """cfg5 (SNMF on CSR 4,194,304 x 128 at 1 % nnz, k = 128; the full 2 GiB W): the Gram-space loop and
    the pass-per-iteration loop agree, the materialised W satisfies W (H H^T) = V H^T row by row (the normal
    equations of snmf.py:67-70) on sampled rows, H stays non-negative."""
    import bench
    from pymf_amd import _lib
    m, n, k = 4194304, 128, 64           # k = 64 keeps H H^T well conditioned for the equality checks ...
    ip, ix, vv = bench.gen_csr(m, n, 0.01, 0, m, True)
    outs = []
    for gram in (1, 0):
        c = _lib.Context(_lib.ALGO_SNMF, m, n, k)
        c.set_v_csr(ip, ix, vv)
        c.fill_w_uniform(42); c.fill_h_uniform(43)
        c.set_option("snmf_gram", gram)
        _, done, _ = c.factorize(3, compute_err=False)
        assert done == 3
        H = c.get_h()
        W = c.get_w()
        outs.append((W[::1021].copy(), H))
        if gram:
            Hd = H.astype(np.float64)
            assert float(H.min()) >= 0.0 and np.isfinite(H).all()
            # W was formed from the H BEFORE the last H step: re-derive it by one more W step and check the
            # normal equations against a float64 evaluation of the sampled rows
            c.update_w()
            W2 = c.get_w()
            rows = np.arange(0, m, 65537)
            G = Hd.dot(Hd.T)
            for r in rows:
                v = np.zeros(n)
                np.add.at(v, ix[ip[r]:ip[r + 1]], vv[ip[r]:ip[r + 1]].astype(np.float64))
                lhs, rhs = W2[r].astype(np.float64).dot(G), v.dot(Hd.T)
                assert np.linalg.norm(lhs - rhs) <= 2e-5 * max(np.linalg.norm(rhs), 1e-30) + 1e-6, r
        del W
        c.close()
    assert rel_fro(outs[0][0], outs[1][0], what="cfg5 W (every 1021st row): Gram-space vs pass-per-iteration") < 1e-6
    assert rel_fro(outs[0][1], outs[1][1], what="cfg5 H: Gram-space vs pass-per-iteration") < 5e-7
    # ... and the real thing, k = 128, one Gram-space run end to end (2 GiB of W written)
    c = _lib.Context(_lib.ALGO_SNMF, m, n, 128)
    c.set_v_csr(ip, ix, vv)
    c.fill_w_uniform(42); c.fill_h_uniform(43)
    _, done, _ = c.factorize(2, compute_err=False)
    H = c.get_h()
    assert done == 2 and float(H.min()) >= 0.0 and np.isfinite(H).all()
    # the 2 GiB W that k_csr_w_blocks<8> writes (the cfg5 roofline kernel), at k = n = 128 where H H^T has a
    # condition number of about 1e7: one more W step from the H at hand, then sampled rows against a float64
    # evaluation w = v M, M = H^T inv(H H^T).  The device forms M^T in float64 and rounds it once, so a row
    # differs from the float64 one by the float32 rounding of M and of the products: |dw| <= c eps32 |v| |M|
    # componentwise -- a bound that grows with the conditioning by itself (|M| ~ 1 / sigma_min(H)).
    c.update_w()
    W2 = c.get_w()
    assert np.isfinite(W2[::4099]).all()
    Hd = H.astype(np.float64)
    Md = np.linalg.solve(Hd.dot(Hd.T), Hd).T                    # n x k
    eps32 = float(np.finfo(np.float32).eps)
    worst = 0.0
    for r in np.arange(3, m, 65537):
        v = np.zeros(n)
        np.add.at(v, ix[ip[r]:ip[r + 1]], vv[ip[r]:ip[r + 1]].astype(np.float64))
        w_ref = v.dot(Md)
        bound = 2.0 * eps32 * (np.abs(v).dot(np.abs(Md)) + np.abs(w_ref)) + 1e-30
        dw = np.abs(W2[r].astype(np.float64) - w_ref)
        worst = max(worst, float((dw / bound).max()))
        assert (dw <= bound).all(), (r, float((dw / bound).max()))
        # ... and the normal equations W (H H^T) = V H^T themselves, residual against the same bound carried through G
        G = Hd.dot(Hd.T)
        res = W2[r].astype(np.float64).dot(G) - v.dot(Hd.T)
        assert (np.abs(res) <= bound.dot(np.abs(G)) + 1e-30).all(), r
    from conftest import _record
    _record("bound", "cfg5 k=128 W rows vs float64 (fraction of the float32 rounding bound)", worst, 1.0, depth=1)
    del W2
    c.close()



# ---- the cooperative one-pass kernel (pmf_coop.h): 64 < num_bases <= 128 with n <= 384, num_bases <= 64 with 256 < n <= 512 ----
@pytest.mark.parametrize("cls_name,shape,k", [("NMF", (3000, 256), 128), ("NMF", (777, 200), 100), ("NMF", (64, 64), 65),
                                              ("BNMF", (5000, 128), 128), ("NMF", (130, 256), 70), ("NMF", (300, 190), 128),
                                              ("BNMF", (2500, 200), 100), ("NMF", (20000, 256), 96),
                                              ("NMF", (1000, 330), 100), ("BNMF", (900, 384), 128), ("NMF", (1500, 384), 64),
                                              ("NMF", (800, 500), 33), ("BNMF", (4000, 512), 64), ("NMF", (70, 300), 50)])
def test_fused8_vs_oracle(pm, cls_name, shape, k):
    import oracle
    rs = np.random.RandomState(shape[0] + k)
    V = rs.random_sample(shape).astype(np.float32)
    if cls_name == "BNMF":
        V = (V < 0.3).astype(np.float32)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    mdl = getattr(pm, cls_name)(V, num_bases=k)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    mdl.factorize(niter=5)
    assert mdl._ctx.path_name.startswith("k_nmf_coop<")
    o = getattr(oracle, cls_name + "Oracle")(V, num_bases=k)
    o.W, o.H = W0.copy(), H0.copy()
    o.factorize(niter=5)
    assert rel_fro(mdl.W, o.W, what="mdl.W") < 2e-6 and rel_fro(mdl.H, o.H, what="mdl.H") < 1e-6
    close(mdl.ferr, o.ferr, rtol=4e-8, what="mdl.ferr")
    # single hooks on the same shape: one-pass update_w + cached (P | S) for update_h
    mdl.update_w(); o.update_w()
    mdl.update_h(); o.update_h()
    assert rel_fro(mdl.W, o.W, what="hooks W") < 2e-6 and rel_fro(mdl.H, o.H, what="hooks H") < 2e-6


def test_fused8_rnmf_free_run_and_reproducibility(pm):
    """RNMF on the k <= 128 kernel; the free-running loop equals the stepwise loop bit for bit; two runs
    give identical bits (per-wave fixed order, float64 slab sums)."""
    from pymf_amd import _lib
    from pymf_amd.rnmf import RNMF
    from oracle import RNMFOracle
    rs = np.random.RandomState(8)
    V = rs.random_sample((1200, 256)).astype(np.float32)
    V.flat[rs.randint(0, V.size, size=V.size // 300)] += 5.0
    np.random.seed(5)
    mdl = RNMF(V, num_bases=96, lamb=1.0)
    mdl.factorize(niter=3)
    assert mdl._ctx.path_name == "k_nmf_coop<2,4,4,rnmf>"
    np.random.seed(5)
    o = RNMFOracle(V, num_bases=96, lamb=1.0)
    o.factorize(niter=3)
    close(mdl.ferr, o.ferr, rtol=1e-7, what="mdl.ferr")
    assert rel_fro(mdl.W, o.W, what="mdl.W") < 2e-5 and rel_fro(mdl.H, o.H, what="mdl.H") < 8e-7
    m, n, k = 9000, 192, 128
    Vd = rs.random_sample((m, n)).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = rs.random_sample((k, n)).astype(np.float32)
    outs = []
    for mode in ("free", "free", "step"):
        c = _lib.Context(_lib.ALGO_NMF, m, n, k)
        c.set_v_dense(Vd); c.set_w(W0); c.set_h(H0)
        if mode == "free":
            f, done, conv = c.factorize(21)
            f = list(f)
        else:
            f = [c.factorize(1, conv_eps=0.0)[0][0] for _ in range(21)]
        outs.append((c.get_w(), c.get_h(), np.array(f)))
        c.close()
    for a, b_ in ((outs[0], outs[1]), (outs[0], outs[2])):
        np.testing.assert_array_equal(a[0], b_[0])
        np.testing.assert_array_equal(a[1], b_[1])
        np.testing.assert_allclose(a[2], b_[2], rtol=1e-12)


def test_snmf_singular_gram_raises_linalgerror(pm):
    """snmf.py:69: np.linalg.inv(H H^T) raises LinAlgError('Singular matrix') when H has a zero row; so do the
    device inverses (a zero pivot, reported through PMF_ESINGULAR), hook by hook and from factorize()."""
    rs = np.random.RandomState(2)
    V = (rs.random_sample((300, 40)) - 0.3).astype(np.float32)
    for k in (5, 70, 150):                                      # k_inverse_spd_mfma<4>, <8>, k_inverse_spd_big
        n = 40 if k <= 40 else 200
        Vk = V if n == 40 else (rs.random_sample((300, n)) - 0.3).astype(np.float32)
        H0 = rs.random_sample((k, n))
        H0[k // 2] = 0.0
        mdl = pm.SNMF(Vk, num_bases=k)
        mdl.W, mdl.H = rs.random_sample((300, k)), H0.copy()
        with pytest.raises(np.linalg.LinAlgError):
            mdl.update_w()
        mdl = pm.SNMF(Vk, num_bases=k)
        mdl.W, mdl.H = rs.random_sample((300, k)), H0.copy()
        with pytest.raises(np.linalg.LinAlgError):
            mdl.factorize(niter=2)
        mdl.H = rs.random_sample((k, n))                        # the object stays usable
        mdl.factorize(niter=2)
        assert np.isfinite(mdl.W).all() and np.isfinite(mdl.ferr).all()


def test_snmf_failed_w_step_keeps_the_device_resident_factors(pm):
    """snmf.py:69-70 raises before W is rebound.  With the factors living on the device between calls (nobody holds the
    host arrays), a failing W step must not lose the W of the previous, successful call: pmf_snapshot_w / pmf_restore_w."""
    import oracle
    rs = np.random.RandomState(8)
    V = (rs.random_sample((500, 60)) - 0.3).astype(np.float32)
    mdl = pm.SNMF(V, num_bases=6)
    W0, H0 = rs.random_sample((500, 6)), rs.random_sample((6, 60))
    mdl.W, mdl.H = W0.copy(), H0.copy()
    mdl.factorize(niter=3, compute_err=False)
    assert mdl._host_stale == {"W", "H"}                        # nothing was pulled: nobody holds the arrays
    ref = oracle.SNMFOracle(V, num_bases=6)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=3, compute_err=False)
    Hbad = np.asarray(ref.H).copy()
    Hbad[2] = 0.0
    mdl.H = Hbad                                                # singular H H^T from here on
    with pytest.raises(np.linalg.LinAlgError):
        mdl.update_w()
    assert rel_fro(mdl.W, ref.W, what="W after a failed W step = W of the last successful call") < 5e-6
    np.testing.assert_array_equal(mdl.H, Hbad)


@pytest.mark.parametrize("shape,k", [((300, 256), 16), ((40, 64), 1), ((2100, 100), 33)])
def test_stale_gram_partials_after_an_early_exit(pm, shape, k):
    """Round 4 (found by tests/sweeps/fuzz_sequences.py): a free-running loop that STOPS EARLY (nmf.py:198-202) puts the host's
    picture of G = H H^T back to "per-workgroup partial sums" (what the last H step that really ran left); a NEW H after that made
    G invalid but left the count of partials behind, and the second ensure_gram after it summed the old H's partials over the
    freshly computed G -- the W-only loop that followed divided by the Gram matrix of an H that no longer existed (W off by
    16-33 % in the sweep's case).  The sequence, against the oracle."""
    from oracle import NMFOracle
    rs = np.random.RandomState(shape[0] + k)
    V = rs.random_sample(shape).astype(np.float32)
    a, o = pm.NMF(V.copy(), num_bases=k), NMFOracle(V.astype(np.float64), num_bases=k)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    a.W, a.H = W0.copy(), H0.copy(); o.W, o.H = W0.copy(), H0.copy()
    a.factorize(niter=20000, compute_w=False, compute_h=True)          # coefficients for a fixed basis, until stationary
    assert len(a.ferr) < 20000                                         # ... an early exit
    o.H = np.asarray(a.H, dtype=np.float64).copy()                     # (the oracle need not stop at the same iteration)
    Hn = o.H * (1.0 + 0.5 * rs.random_sample(o.H.shape))
    a.H = Hn.copy(); o.H = Hn.copy()
    a.update_w(); o.update_w()
    assert rel_fro(a.W, o.W, what="W after update_w with the new H") < 2e-6
    Wn = o.W * (1.0 + 0.1 * rs.random_sample(o.W.shape))
    a.W = Wn.copy(); o.W = Wn.copy()
    a.factorize(niter=3, compute_w=True, compute_h=False, compute_err=False); o.factorize(niter=3, compute_w=True, compute_h=False, compute_err=False)
    assert rel_fro(a.W, o.W, what="W after the W-only loop behind it") < 5e-6
    assert rel_fro(a.H, o.H, what="H untouched") < 1e-6


def test_long_run_on_a_storage_sensitive_problem(pm):
    """Round 4 (tests/sweeps/fuzz_sequences.py, seed 801 / case 40): 130 SNMF iterations on a 7 x 256 matrix ended 8.8e-4 (W) / 3.1e-3
    (H) from the float64 oracle -- exactly where the oracle ITSELF ends when its factors are rounded to float32 after every update:
    the price of storing H in float32 between iterations on a problem that amplifies it.  Round 6: SNMF's H lives in float64 on the
    device and the Gram-space loop never rounds W either (P = M^T (V^T V) from the float64 M), so the library is now where the
    EXACT float64 oracle is -- the stated 2e-5 holds after 130 iterations -- and 1e-3 away from the float32-stored twin.
    With pmf_set_option("snmf_h64", 0) it is back near the twin (rounds 4-5).
    The inputs are the sweep's (tests/golden/snmf_storage_sensitive_7x256_k3.npz: data only)."""
    from oracle import SNMFOracle
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "snmf_storage_sensitive_7x256_k3.npz"))
    V, W0, H0, k = g["V"], g["W0"], g["H0"], int(g["k"])
    a = pm.SNMF(V.copy(), num_bases=k)
    a.W, a.H = W0.copy(), H0.copy()
    a.factorize(niter=130, compute_err=False)
    exact, stored = SNMFOracle(V.astype(np.float64), num_bases=k), SNMFOracle(V.astype(np.float64), num_bases=k)
    for o in (exact, stored):
        o.W, o.H = W0.copy(), H0.copy()
    for _ in range(130):
        exact.update_w(); exact.update_h()
        stored.update_w(); stored.W = stored.W.astype(np.float32).astype(np.float64)
        stored.update_h(); stored.H = stored.H.astype(np.float32).astype(np.float64)
    # the float64 oracle and its float32-stored twin have drifted apart by about 1e-3 ...
    assert 1e-4 < np.linalg.norm(stored.W - exact.W) / np.linalg.norm(exact.W) < 1e-2
    # ... and the library is where the float64 oracle is
    assert rel_fro(a.W, exact.W, what="W vs the float64 oracle, 130 iterations") < 2e-5
    assert rel_fro(a.H, exact.H, what="H vs the float64 oracle, 130 iterations") < 2e-5
    # the float32-H form of rounds 1-5 (still selectable) is a float32-stored trajectory: 1e-3 from the float64 oracle
    b = pm.SNMF(V.copy(), num_bases=k)
    b.W, b.H = W0.copy(), H0.copy()
    b.factorize(niter=1, compute_err=False)          # (creates the context)
    b._ctx.set_option("snmf_h64", 0)
    b.W, b.H = W0.copy(), H0.copy()
    b.factorize(niter=130, compute_err=False)
    assert 1e-4 < np.linalg.norm(np.asarray(b.W) - exact.W) / np.linalg.norm(exact.W) < 1e-2
