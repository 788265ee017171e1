"""GPU parity for NMFALS (batched exact active-set QP) and SNMF on scipy.sparse CSR data."""
import numpy as np
import pytest

from conftest import load_golden, rel_fro, close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pm():
    import pymf_amd
    from pymf_amd import _lib
    assert _lib.device_count() >= 1
    return pymf_amd


@pytest.mark.parametrize("name", ["nnls_24x18_k4", "nnls_reftest"])
def test_nmfals_vs_reference_nnls_golden(pm, name):
    """The reference's NNLS sibling (pymf/nmfnnls.py) minimises the same objective as
    NMFALS' cvxopt QP (nmfals.py:74,89); its golden pins the exact minimiser."""
    g = load_golden(name)
    mdl = pm.NMFALS(g["V"], num_bases=int(g["k"]))
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert len(mdl.ferr) == len(g["ferr"])
    assert (mdl.W >= 0).all() and (mdl.H >= 0).all()
    if name == "nnls_reftest":
        # tests/test_pymf.py:32-33 data is rank 3 and k = 4: the Gram matrices are singular to
        # working precision, the factors are not unique -- compare what IS determined: the
        # reconstruction, the error curve and the reference test's own bound (:86-88).
        assert rel_fro(mdl.W.dot(mdl.H), g["W"].dot(g["H"]), what="mdl.W.dot(mdl.H)") < 6e-5
        close(mdl.ferr, g["ferr"], rtol=2e-2, atol=1e-4, what="mdl.ferr")
        assert mdl.ferr[-1] / (g["V"].shape[0] + g["V"].shape[1]) < 0.1
        return
    assert np.max(np.abs(mdl.W - g["W"])) < 5e-5 * max(1.0, np.abs(g["W"]).max())
    assert np.max(np.abs(mdl.H - g["H"])) < 5e-5 * max(1.0, np.abs(g["H"]).max())
    close(mdl.ferr, g["ferr"], rtol=2e-4, atol=1e-6, what="mdl.ferr")


@pytest.mark.parametrize("cls_name", ["NMFALS", "NMFNNLS"])
@pytest.mark.parametrize("name", ["nnls_130x90_k33", "nnls_cfg3s", "nnls_300x200_k72", "nnls_260x300_k130"])
def test_nmfals_vs_the_real_nmfnnls_reference_at_cfg3_class_shapes(pm, name, cls_name):
    """Round 4: goldens from pymf/nmfnnls.py -- unmodified reference arithmetic, REAL scipy.optimize.nnls,
    the objective of nmfals.py:74,89 -- at cfg3's width and num_bases (2048 x 1024, k = 64), at 33 bases, and beyond 64 and
    128 bases (k_nnqp_wave / k_nnqp_big).  Tolerance: DESIGN.md section 4 (fp32 V H^T and fp32 storage of the factors
    through the conditioning of the k x k systems), next to SURVEY 8(c)'s float64 figure."""
    g = load_golden(name)
    mdl = getattr(pm, cls_name)(g["V"], num_bases=int(g["k"]))
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert len(mdl.ferr) == len(g["ferr"])
    assert (mdl.W >= 0).all() and (mdl.H >= 0).all()
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 1e-4
    assert rel_fro(mdl.H, g["H"], what="mdl.H") < 1e-4
    close(mdl.ferr, g["ferr"], rtol=6e-7, what="mdl.ferr")
    # supports: a variable the reference holds at exactly zero may come out as a tiny positive only
    big = g["W"] > 1e-3 * g["W"].max()
    assert (mdl.W[big] > 0).all()
    assert np.abs(mdl.W[g["W"] == 0]).max(initial=0.0) < 5e-4 * g["W"].max()


@pytest.mark.parametrize("name", ["nmfals_24x18_k4", "nmfals_130x90_k33", "nmfals_cfg3s", "nmfals_reftest"])
def test_nmfals_data_flow_pin_vs_nmfals_py_with_stub_cvxopt(pm, name):
    """DATA-FLOW pins: pymf/nmfals.py itself (nmfals.py:70-97) driven by an exact-QP stand-in for the absent cvxopt
    (tests/golden/gen_golden.py) -- signs, float64 casts, column / row scatter, eager map.  The NUMBERS are pinned by the
    nnls_* fixtures above (real reference arithmetic); nmfals_cfg3s has cfg3's width and num_bases."""
    g = load_golden(name)
    mdl = pm.NMFALS(g["V"], num_bases=int(g["k"]))
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert len(mdl.ferr) == len(g["ferr"])
    assert (mdl.W >= 0).all() and (mdl.H >= 0).all()
    if name == "nmfals_reftest":           # singular Gram matrices: only W H and the error curve are determined
        assert rel_fro(mdl.W.dot(mdl.H), g["W"].dot(g["H"]), what="mdl.W.dot(mdl.H)") < 6e-5
        close(mdl.ferr, g["ferr"], rtol=2e-2, atol=1e-4, what="mdl.ferr")
        return
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 1e-4
    assert rel_fro(mdl.H, g["H"], what="mdl.H") < 8e-5
    close(mdl.ferr, g["ferr"], rtol=2e-7, what="mdl.ferr")


def test_nmfnnls_class_matches_reference_golden(pm):
    """pymf.NMFNNLS (nmfnnls.py:69-80) is served by the same kernel; golden = the reference itself."""
    g = load_golden("nnls_24x18_k4")
    mdl = pm.NMFNNLS(g["V"], num_bases=int(g["k"]))
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert np.max(np.abs(mdl.W - g["W"])) < 5e-5 * max(1.0, np.abs(g["W"]).max())
    assert np.max(np.abs(mdl.H - g["H"])) < 5e-5 * max(1.0, np.abs(g["H"]).max())
    close(mdl.ferr, g["ferr"], rtol=2e-4, atol=1e-6, what="mdl.ferr")


@pytest.mark.parametrize("m,n,k", [(40, 30, 4), (70, 50, 16), (130, 90, 33), (96, 64, 64)])
def test_nmfals_vs_oracle(pm, m, n, k):
    from oracle import NMFALSOracle
    rs = np.random.RandomState(m + k)
    V = rs.random_sample((m, n)).astype(np.float32)
    W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
    mdl = pm.NMFALS(V, num_bases=k)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    mdl.factorize(niter=2)
    ref = NMFALSOracle(V, num_bases=k)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=2)
    # exact QP minimisers on both sides; the device forms the right-hand sides in float32
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 3e-5
    assert rel_fro(mdl.H, ref.H, what="mdl.H") < 4e-5
    close(mdl.ferr, ref.ferr, rtol=3e-7, what="mdl.ferr")
    assert (mdl.W >= 0).all() and (mdl.H >= 0).all()


def test_nmfals_kkt_property_large(pm):
    """Size-independent property: after update_w every row satisfies the KKT conditions of
    its QP: x >= 0, g = HA x - f >= -tol, and x . g ~ 0."""
    rs = np.random.RandomState(9)
    m, n, k = 20000, 256, 64
    V = rs.random_sample((m, n)).astype(np.float32)
    H = rs.random_sample((k, n))
    mdl = pm.NMFALS(V, num_bases=k)
    mdl.W, mdl.H = np.zeros((m, k)), H.copy()
    mdl.update_w()
    HA = H.dot(H.T)
    F = V.astype(np.float64).dot(H.T)
    g = mdl.W.dot(HA) - F
    scale = np.abs(F).max()
    assert (mdl.W >= 0).all()
    assert g.min() > -2e-4 * scale
    assert np.abs(mdl.W * g).max() < 2e-4 * scale * max(1.0, mdl.W.max())


def test_snmf_csr_matches_dense_oracle(pm):
    import scipy.sparse as sp
    from oracle import SNMFOracle
    g = load_golden("snmf_sparse1pct")
    Vd = g["V"]
    mdl = pm.SNMF(sp.csr_matrix(Vd), num_bases=int(g["k"]))
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]), compute_err=False)
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 1e-5      # reference SNMF on V.toarray()
    assert rel_fro(mdl.H, g["H"], what="mdl.H") < 4e-6
    assert mdl.frobenius_norm() == -123456     # nmf.py:109-112 sentinel for sparse data
    with pytest.raises(TypeError):
        mdl.factorize(niter=2)                 # compute_err=True is meaningless on sparse data
    # a wider, mixed-sign case against the oracle
    rs = np.random.RandomState(3)
    m, n, k = 3000, 128, 128
    Vs = sp.random(m, n, density=0.01, format="csr", dtype=np.float32, random_state=rs)
    W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n)) + 0.1
    a = pm.SNMF(Vs, num_bases=k)
    a.W, a.H = W0.copy(), H0.copy()
    a.factorize(niter=2, compute_err=False)
    ref = SNMFOracle(Vs.toarray(), num_bases=k)
    ref.W, ref.H = W0.copy(), H0.copy()
    ref.factorize(niter=2, compute_err=False)
    assert rel_fro(a.W, ref.W, what="a.W") < 8e-5          # inv(H H^T) at k = n = 128 is ill-conditioned
    assert rel_fro(a.H, ref.H, what="a.H") < 3e-7


def test_nmf_rejects_sparse(pm):
    import scipy.sparse as sp
    mdl = pm.NMF(sp.csr_matrix(np.eye(8, dtype=np.float32)), num_bases=2)
    with pytest.raises(TypeError):
        mdl.factorize(niter=1, compute_err=False)


# ---- k_nnqp_quad (pmf_nnls_quad.h): sixteen lanes per problem, block principal pivoting ----------------------------
@pytest.mark.parametrize("shape,k", [((20000, 300), 50), ((17000, 130), 64), ((18000, 90), 20), ((16500, 64), 7), ((3000, 200), 40)])
def test_nnqp_quad_gives_the_minimisers_of_the_lane_per_variable_kernel(pm, shape, k):
    """The QPs are strictly convex: the minimiser is unique whatever path an exact active-set method takes.  Forced on
    (`nnqp_quad` = 2: every half step, small ones too) against off (k_nnqp), three ALS iterations from the same start;
    then against the float64 oracle."""
    import oracle
    from pymf_amd import _lib
    m, n = shape
    rs = np.random.RandomState(m + k)
    V = rs.random_sample((m, n)).astype(np.float32)
    V[rs.random_sample((m, n)) < 0.3] = 0.0
    W0, H0 = rs.random_sample((m, k)).astype(np.float32), rs.random_sample((k, n)).astype(np.float32)
    out = {}
    for quad in (2, 0):
        c = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
        c.set_option("nnqp_quad", quad)
        c.set_v_dense(V); c.set_w(W0); c.set_h(H0)
        ferr, done, _ = c.factorize(3, compute_err=True)
        out[quad] = (c.get_w(), c.get_h(), ferr)
        c.close()
    assert rel_fro(out[2][0], out[0][0], what="W: k_nnqp_quad vs k_nnqp") < 2e-5
    assert rel_fro(out[2][1], out[0][1], what="H: k_nnqp_quad vs k_nnqp") < 2e-5
    close(out[2][2], out[0][2], rtol=1e-6, what="ferr: k_nnqp_quad vs k_nnqp")
    assert float(out[2][0].min()) >= 0.0 and float(out[2][1].min()) >= 0.0
    if m <= 3000:
        o = oracle.NMFALSOracle(V, num_bases=k)
        o.W, o.H = W0.astype(np.float64), H0.astype(np.float64)
        o.factorize(niter=3)
        assert rel_fro(out[2][0], o.W, what="W: k_nnqp_quad vs the float64 oracle") < 1e-4
        assert rel_fro(out[2][1], o.H, what="H: k_nnqp_quad vs the float64 oracle") < 1e-4


def test_nnqp_quad_leaves_dead_bases_at_zero(pm):
    """A basis that has died out (zero row of H: zero row and column of HA = H H^T) never becomes passive; the kernel
    patches it to the identity before inverting HA (k_nnqp_patch_dead) and k_spd_unique still lets the warm start through."""
    from pymf_amd import _lib
    rs = np.random.RandomState(5)
    m, n, k = 17000, 120, 48
    V = rs.random_sample((m, n)).astype(np.float32)
    W0, H0 = rs.random_sample((m, k)).astype(np.float32), rs.random_sample((k, n)).astype(np.float32)
    H0[11] = 0.0
    res = {}
    for quad in (2, 0):
        c = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
        c.set_option("nnqp_quad", quad)
        c.set_v_dense(V); c.set_w(W0); c.set_h(H0)
        c.update_w()
        res[quad] = c.get_w()
        c.close()
    assert np.all(res[2][:, 11] == 0.0) and np.isfinite(res[2]).all()
    assert rel_fro(res[2], res[0], what="W with a dead basis: k_nnqp_quad vs k_nnqp") < 2e-5


@pytest.mark.parametrize("shape,k,niter", [((20000, 300), 50, 8), ((32768, 512), 64, 12), ((18000, 90), 20, 6)])
def test_nnqp_quad_frames_are_bit_identical(pm, shape, k, niter):
    """`nnqp_frame16`: the 16-slot frame (three waves per SIMD) with the 32-slot frame behind it for the problems that
    outgrow it, or the 32-slot frame for all -- a problem's unknowns sit right-aligned in either frame, in the same lanes,
    so the arithmetic is the same: W and H bit for bit, over iterations that start with every system beyond 16 unknowns
    (random start: the device-side switch hands whole half steps to the 32-slot frame) and end with most of them inside."""
    from pymf_amd import _lib
    m, n = shape
    out = []
    for f16 in (1, 0):
        c = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
        c.set_option("nnqp_quad", 2)
        c.set_option("nnqp_frame16", f16)
        c.fill_v_uniform(1234); c.fill_w_uniform(42); c.fill_h_uniform(43)
        c.factorize(niter, compute_err=False)
        out.append((c.get_w(), c.get_h()))
        c.close()
    assert np.isfinite(out[0][0]).all() and np.isfinite(out[0][1]).all()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])


def test_snmf_csr_shape_beyond_the_lds_accumulator(pm):
    """Round 4 (tests/sweeps/fuzz_sequences.py): CSR data whose n x num_bases image does not fit the 160 KiB LDS accumulator of the
    CSR scatter (520 columns x 100 bases) used to fail in update_h ("n * num_bases too large for the LDS accumulator"); the rows are
    expanded once on the device instead, as for > 128 bases.  Against the oracle on the dense matrix."""
    import scipy.sparse as sp
    from oracle import SNMFOracle
    rs = np.random.RandomState(9)
    V = ((rs.random_sample((600, 520)) - 0.4) * (rs.random_sample((600, 520)) < 0.2)).astype(np.float32)
    k = 100
    W0, H0 = rs.random_sample((600, k)), rs.random_sample((k, 520))
    a, o = pm.SNMF(sp.csr_matrix(V), num_bases=k), SNMFOracle(V.astype(np.float64), num_bases=k)
    a.W, a.H = W0.copy(), H0.copy(); o.W, o.H = W0.copy(), H0.copy()
    a.update_w(); o.update_w(); a.update_h(); o.update_h()
    a.factorize(niter=2, compute_err=False); o.factorize(niter=2, compute_err=False)
    assert rel_fro(a.W, o.W, what="mdl.W") < 2e-4 and rel_fro(a.H, o.H, what="mdl.H") < 2e-5


@pytest.mark.parametrize("shape,k,niter", [((32768, 1024), 64, 5), ((20000, 512), 40, 6), ((3000, 2048), 64, 4), ((17000, 768), 50, 5)])
def test_fused_k_by_k_chain_gives_the_bits_of_the_two_launch_chain(pm, shape, k, niter):
    """Round 6 ('3b'): at 49-64 bases the k x k chain of an NMFALS half step CAN run as one launch (option
    fuse_chain; off by default: it measured 1-2 % slower than two launches) -- the workgroup
    that completes H H^T (k_gram_splitk<float, true>) resp. the slab sum with the Hessian W^T W (k_reduce_slabs_inv) goes on to
    invert it: uniqueness flag, patched Hessian and B = inv(HA) for the QP kernels.  Same sums in the same order and the same
    inversion body as the two launches before (option fuse_chain = 0): W, H and ferr must be bit-identical, hook by hook too."""
    from pymf_amd import _lib
    rs = np.random.RandomState(shape[1] + k)
    V = rs.random_sample(shape).astype(np.float32)
    W0 = rs.random_sample((shape[0], k)).astype(np.float32)
    H0 = rs.random_sample((k, shape[1])).astype(np.float32)
    outs = []
    for fuse in (3, 0):          # (both half steps fused / the default: two launches each -- measured faster, profiles/r06_experiments.md)
        c = _lib.Context(_lib.ALGO_NMFALS, shape[0], shape[1], k)
        c.set_v_dense(V); c.set_w(W0); c.set_h(H0)
        c.set_option("fuse_chain", fuse)
        ferr, done, _ = c.factorize(niter, compute_err=True)
        assert done == niter
        a = (c.get_w(), c.get_h(), ferr.copy())
        c.update_w(); c.update_h()
        outs.append(a + (c.get_w(), c.get_h()))
        c.close()
    for x, y in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(x, y)
    assert (outs[0][0] >= 0).all() and np.isfinite(outs[0][2]).all()
