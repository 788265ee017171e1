"""GPU parity beyond the register-resident base counts: SNMF / RNMF / NNDSVD with num_bases > 128 and
NMFALS / NMFNNLS with num_bases > 64 (the reference has no limit: snmf.py:69-70, nmfals.py:78-80,
rnmf.py:109-115, nndsvd.py:92-106).  These widths run the generic paths -- base blocks of 128 on the
tiled kernels, the cooperative float64 inverse (k_inverse_spd_big), the active-set QP with its inverse
images in global memory (k_nnqp_big) -- against goldens produced by the reference itself
(tests/golden/gen_golden.py, `bigk_*`) and the float64 oracles."""
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


def _run_golden(pm, cls_name, name):
    g = load_golden(name)
    mdl = getattr(pm, cls_name)(g["V"], num_bases=int(g["k"]))
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert len(mdl.ferr) == len(g["ferr"])
    return g, mdl


def test_snmf_k160_vs_reference_golden(pm):
    g, mdl = _run_golden(pm, "SNMF", "bigk_snmf_512x320_k160")
    close(mdl.ferr, g["ferr"], rtol=7e-8, what="mdl.ferr")
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 2e-6 and rel_fro(mdl.H, g["H"], what="mdl.H") < 6e-7


@pytest.mark.parametrize("name", ["bigk_nmfals_300x200_k72", "bigk_nmfals_260x300_k130"])
def test_nmfals_beyond_64_data_flow_pin_with_stub_cvxopt(pm, name):
    """DATA-FLOW pin: nmfals.py itself (exact-QP stand-in for cvxopt, gen_golden.load_reference_nmfals); k = 130 also
    crosses the 128-base block boundary of the products around the QPs.  Numbers: the nnls_300x200_k72 /
    nnls_260x300_k130 fixtures (real pymf/nmfnnls.py) in tests/test_gpu_als_sparse.py."""
    g, mdl = _run_golden(pm, "NMFALS", name)
    close(mdl.ferr, g["ferr"], rtol=6e-7, what="mdl.ferr")
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 9e-5 and rel_fro(mdl.H, g["H"], what="mdl.H") < 1e-4


def test_rnmf_k140_vs_reference_golden(pm):
    from pymf_amd.rnmf import RNMF
    g = load_golden("bigk_rnmf_300x256_k140")
    np.random.seed(int(g["seed"]))
    mdl = RNMF(g["V"], num_bases=int(g["k"]), lamb=float(g["lamb"]))
    mdl.factorize(niter=int(g["niter"]))
    close(mdl.ferr, g["ferr"], rtol=3e-7, what="mdl.ferr")
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 2e-5 and rel_fro(mdl.H, g["H"], what="mdl.H") < 2e-6
    assert rel_fro(mdl.S, g["S"], what="mdl.S") < 2e-6


def test_nndsvd_k150_vs_reference_golden(pm):
    g = load_golden("bigk_nndsvd_500x300_k150")
    mdl = pm.NNDSVD(g["V"], num_bases=int(g["k"]))
    mdl.factorize()
    # the trailing singular directions of a random matrix are close together: their vectors carry the float32
    # Gram matrix's rounding amplified by 1 / gap (the reference forms the same product in float32 too)
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 3e-4 and rel_fro(mdl.H, g["H"], what="mdl.H") < 3e-4
    close(mdl.ferr, g["ferr"], rtol=4e-6, what="mdl.ferr")


@pytest.mark.parametrize("shape,k,mode", [((3000, 400), 160, "loop"), ((3000, 400), 160, "pass"), ((3000, 400), 160, "hooks"),
                                          ((2000, 700), 300, "loop"), ((1500, 1200), 520, "loop"), ((900, 1100), 1024, "loop")])
def test_snmf_wide_vs_float64_oracle(pm, shape, k, mode):
    """loop: factorize() (Gram space); pass: snmf_gram = 0, one pass over V per iteration; hooks: update_w /
    update_h called one by one."""
    from oracle import SNMFOracle
    rs = np.random.RandomState(sum(shape) + k)
    V = (rs.random_sample(shape) - 0.4).astype(np.float32)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    o = SNMFOracle(V.astype(np.float64), num_bases=k)
    o.W, o.H = W0.copy(), H0.copy()
    mdl = pm.SNMF(V, num_bases=k)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    if mode == "hooks":
        for _ in range(3):
            mdl.update_w(); o.update_w()
            mdl.update_h(); o.update_h()
        assert abs(mdl.frobenius_norm() - o.frobenius_norm()) <= 1e-6 * o.frobenius_norm()
    else:
        if mode == "pass":
            mdl._context().set_option("snmf_gram", 0)
        mdl.factorize(niter=3)
        o.factorize(niter=3)
        # k = 1024 on 1100 columns: H H^T is nearly singular (k / n = 0.93), the float32 roundings of M^T and
        # (P | S) show in the sixth digit of the error -- 2e-6 or 8e-6 depending on nothing but the order in which
        # the row chunks' partial sums are added (64- vs 16-row chunk granularity); the other shapes sit at 1e-8
        close(mdl.ferr, o.ferr, rtol=3e-5 if k == 1024 else 1e-6, what="mdl.ferr")
    assert rel_fro(mdl.W, o.W, what="mdl.W") < 2e-5 and rel_fro(mdl.H, o.H, what="mdl.H") < 2e-6


@pytest.mark.parametrize("hooks", [False, True])
def test_snmf_wide_on_sparse_data(pm, hooks):
    """scipy.sparse data with more than 128 bases: the rows are expanded once on the device (k_csr_densify) and
    the dense kernels run; semantics = SNMF on data.toarray(), as for the CSR kernels."""
    import scipy.sparse as sp
    from oracle import SNMFOracle
    rs = np.random.RandomState(3)
    Vd = (rs.random_sample((2000, 300)) * (rs.random_sample((2000, 300)) < 0.1)).astype(np.float32)
    W0, H0 = rs.random_sample((2000, 200)), rs.random_sample((200, 300))
    o = SNMFOracle(Vd.astype(np.float64), num_bases=200)
    o.W, o.H = W0.copy(), H0.copy()
    mdl = pm.SNMF(sp.csr_matrix(Vd), num_bases=200)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    if hooks:
        for _ in range(2):
            mdl.update_w(); o.update_w()
            mdl.update_h(); o.update_h()
    else:
        mdl.factorize(niter=3, compute_err=False)
        o.factorize(niter=3, compute_err=False)
    assert mdl.frobenius_norm() == -123456                      # nmf.py:109-112
    assert rel_fro(mdl.W, o.W, what="mdl.W") < 3e-6 and rel_fro(mdl.H, o.H, what="mdl.H") < 6e-7


@pytest.mark.parametrize("cls_name,shape,k,hooks", [("NMFALS", (600, 200), 80, False), ("NMFALS", (500, 300), 100, True),
                                                    ("NMFALS", (400, 380), 200, False), ("NMFNNLS", (300, 260), 96, False),
                                                    ("NMFALS", (150, 700), 520, False), ("NMFALS", (90, 400), 100, False)])
def test_als_wide_vs_float64_oracle(pm, cls_name, shape, k, hooks):
    """k_nnqp_wave (65-128 bases) and k_nnqp_big with 4 and 16 variables per lane; the last two cases have more bases than
    rows (rank-deficient Hessians: the cold-start path, as the reference test's rank-3 data -- at 100 bases k_nnqp_wave
    stands back on the inverse's pivots and k_nnqp_big<2> takes the half step)."""
    import oracle
    rs = np.random.RandomState(sum(shape) + k)
    V = rs.random_sample(shape).astype(np.float32)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    ocls = getattr(oracle, "NMFALSOracle")
    o = ocls(V.astype(np.float64), num_bases=k)
    o.W, o.H = W0.copy(), H0.copy()
    mdl = getattr(pm, cls_name)(V, num_bases=k)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    if hooks:
        for _ in range(2):
            mdl.update_w(); o.update_w()
            mdl.update_h(); o.update_h()
    else:
        mdl.factorize(niter=2)
        o.factorize(niter=2)
        close(mdl.ferr, o.ferr, rtol=1e-6, what="mdl.ferr")
    if k < min(shape):
        assert rel_fro(mdl.W, o.W, what="mdl.W") < 2e-4 and rel_fro(mdl.H, o.H, what="mdl.H") < 2e-4
    # the fit itself is what a degenerate problem pins down
    assert abs(mdl.frobenius_norm() - o.frobenius_norm()) <= 1e-5 * max(o.frobenius_norm(), 1.0)


@pytest.mark.parametrize("wave", [0, 1])
def test_nnqp_big_equals_register_kernel(pm, wave):
    """The kernels beyond 64 bases (k_nnqp_big, one variable at a time; k_nnqp_wave, block principal pivoting) end at
    the KKT point of the register-resident one: on a k = 64 problem solved as-is and embedded in a 65-variable problem
    whose extra variable is a dead basis, the first 64 coordinates agree to rounding."""
    from pymf_amd import _lib
    rs = np.random.RandomState(4)
    m, n = 900, 120
    V = rs.random_sample((m, n)).astype(np.float32)
    outs = []
    for k in (64, 65):
        H = np.zeros((k, n), dtype=np.float32)
        H[:64] = np.random.RandomState(5).random_sample((64, n))
        if k == 65:
            H[64] = 0.0                                      # dead basis: zero row of H H^T, stays out
        ctx = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
        ctx.set_option("nnqp_wave", wave)
        ctx.set_v_dense(V); ctx.set_w(np.zeros((m, k), dtype=np.float32)); ctx.set_h(H)
        ctx.update_w()
        outs.append(ctx.get_w())
        ctx.close()
    assert np.all(outs[1][:, 64] == 0.0)
    assert rel_fro(outs[1][:, :64], outs[0], what="beyond-64 kernel vs k_nnqp") < (1e-6 if wave else 1e-9)


@pytest.mark.parametrize("m,n,k,zero_rows", [(3000, 300, 128, 0), (2000, 260, 72, 0), (1500, 400, 100, 3), (700, 900, 127, 1)])
def test_nnqp_wave_equals_nnqp_big(pm, m, n, k, zero_rows):
    """k_nnqp_wave (pmf_nnls_wave.h) against k_nnqp_big over whole iterations from a cold start and warm: both half
    steps, dead bases (zero rows of H), data with structural zeros.  The QPs are strictly convex: one KKT point."""
    from pymf_amd import _lib
    rs = np.random.RandomState(m + k)
    V = (rs.random_sample((m, n)) * (rs.random_sample((m, n)) < 0.6)).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = rs.random_sample((k, n)).astype(np.float32)
    H0[:zero_rows] = 0.0
    res = []
    for wave in (0, 1):
        ctx = _lib.Context(_lib.ALGO_NMFALS, m, n, k)
        ctx.set_option("nnqp_wave", wave)
        ctx.set_v_dense(V); ctx.set_w(W0); ctx.set_h(H0)
        ferr, done, _ = ctx.factorize(4, compute_err=True, conv_eps=0.0)
        res.append((ctx.get_w(), ctx.get_h(), np.asarray(ferr[:done], dtype=np.float64)))
        ctx.close()
    (Wb, Hb, fb), (Ww, Hw, fw) = res
    close(fw, fb, rtol=2e-6, what="ferr: wave vs big")
    assert rel_fro(Ww, Wb, what="W: wave vs big") < 1e-4 and rel_fro(Hw, Hb, what="H: wave vs big") < 1e-4
    assert Ww.min() >= 0.0 and Hw.min() >= 0.0 and np.isfinite(Ww).all() and np.isfinite(Hw).all()


def test_rnmf_wide_vs_float64_oracle(pm):
    from pymf_amd.rnmf import RNMF
    from oracle import RNMFOracle
    rs = np.random.RandomState(8)
    V = rs.random_sample((700, 520)).astype(np.float32)
    V.flat[rs.randint(0, V.size, size=V.size // 300)] += 5.0
    np.random.seed(5)
    mdl = RNMF(V, num_bases=260, lamb=1.0)
    mdl.factorize(niter=3)
    np.random.seed(5)
    o = RNMFOracle(V, num_bases=260, lamb=1.0)
    o.factorize(niter=3)
    close(mdl.ferr, o.ferr, rtol=9e-8, what="mdl.ferr")
    assert rel_fro(mdl.W, o.W, what="mdl.W") < 3e-5 and rel_fro(mdl.H, o.H, what="mdl.H") < 2e-6
    assert rel_fro(mdl.S, o.S, what="mdl.S") < 3e-6
    # hooks one by one (S exists after factorize)
    mdl.update_w(); o.update_w()
    mdl.update_h(); o.update_h()
    assert rel_fro(mdl.W, o.W, what="hooks W") < 3e-5 and rel_fro(mdl.H, o.H, what="hooks H") < 2e-6


def test_nndsvd_wide_bases_vs_float64_oracle(pm):
    from oracle import nndsvd_closed_form
    rs = np.random.RandomState(11)
    r = 190
    A = rs.random_sample((1500, r)) * (1.0 + np.arange(r))[None, ::-1]
    V = (A @ rs.random_sample((r, 400)) + 0.05 * rs.random_sample((1500, 400))).astype(np.float32)
    mdl = pm.NNDSVD(V, num_bases=180)
    mdl.factorize()
    W, H = nndsvd_closed_form(V, 180)
    assert rel_fro(mdl.W, W, what="mdl.W") < 1e-3 and rel_fro(mdl.H, H, what="mdl.H") < 1e-3
    ref_err = np.linalg.norm(V.astype(np.float64) - W @ H)
    assert abs(mdl.ferr[0] - ref_err) <= 1e-3 * max(ref_err, 1e-3)


@pytest.mark.parametrize("cls_name,shape,k,niter", [("NMF", (3000, 2600), 1500, 3), ("BNMF", (3000, 2600), 1500, 3),
                                                    ("SNMF", (3000, 2600), 1500, 2), ("RNMF", (3000, 2600), 1500, 2),
                                                    ("NMF", (3000, 2800), 2432, 1), ("SNMF", (3000, 2800), 2432, 1)])
def test_beyond_1024_bases_vs_float64_oracle(pm, cls_name, shape, k, niter):
    """The reference has no limit on num_bases (nmf.py:116-120, snmf.py:69-70); round 4 lifted the library's from 1 024 to
    2 432 (NMFALS / NMFNNLS stay at 1 024): the generic kernels against the float64 oracles at 1 500 and at the limit."""
    import oracle
    from pymf_amd.rnmf import RNMF
    from pymf_amd.bnmf import BNMF
    rs = np.random.RandomState(sum(shape) + k)
    V = (rs.random_sample(shape) - (0.5 if cls_name == "SNMF" else 0.0)).astype(np.float32)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    kw = {"lamb": 1.0} if cls_name == "RNMF" else {}
    o = getattr(oracle, cls_name + "Oracle")(V.astype(np.float64), num_bases=k, **kw)
    cls = {"RNMF": RNMF, "BNMF": BNMF}.get(cls_name) or getattr(pm, cls_name)
    mdl = cls(V, num_bases=k, **kw)
    if cls_name == "RNMF":                 # its init_w / init_h are part of the algorithm (rnmf.py:76-96): same seed, same stream
        np.random.seed(5); o.factorize(niter=niter)
        np.random.seed(5); mdl.factorize(niter=niter)
    else:
        o.W, o.H = W0.copy(), H0.copy()
        mdl.W, mdl.H = W0.copy(), H0.copy()
        o.factorize(niter=niter)
        mdl.factorize(niter=niter)
    assert rel_fro(mdl.W, o.W, what="mdl.W") < (2e-4 if cls_name == "RNMF" else 1e-5)
    assert rel_fro(mdl.H, o.H, what="mdl.H") < 5e-6
    close(mdl.ferr, o.ferr, rtol=5e-6, what="mdl.ferr")


def test_num_bases_limits_are_reported(pm):
    with pytest.raises(Exception) as e:
        pm.NMF(np.ones((8, 8), dtype=np.float32), num_bases=2433).factorize(niter=1)
    assert "2432" in str(e.value)
    with pytest.raises(Exception) as e:
        pm.NMFALS(np.ones((8, 8), dtype=np.float32), num_bases=1025).factorize(niter=1)
    assert "1024" in str(e.value)
