"""Pin the CPU oracle (oracle/) against golden vectors produced by the real reference.

The fixtures in tests/golden/*.npz were written by tests/golden/gen_golden.py,
which imports nils-werner/pymf unmodified.  Same BLAS + same op order => the
restatement must reproduce them to ~1e-6 relative (last digits are BLAS-order
dependent; the generator and this test may run on different hosts).
"""
import numpy as np
import pytest

from conftest import load_golden, rel_fro
from oracle import NMFOracle, SNMFOracle, NMFALSOracle, BNMFOracle

ORACLE = {"nmf": NMFOracle, "snmf": SNMFOracle, "nnls": NMFALSOracle, "bnmf": BNMFOracle}

CASES = ["nmf_cfg1_f64", "nmf_cfg1_f32", "snmf_cfg1_f64", "snmf_cfg1_f32",
         "nmf_512x128_k16", "snmf_512x128_k16", "nmf_cfg4s", "snmf_cfg4s",
         "nmf_cfg2s", "snmf_cfg2s", "nmf_cfg5s_dense", "snmf_cfg5s_dense",
         "snmf_sparse1pct", "nmf_37x29_k5", "snmf_37x29_k5", "nmf_reftest",
         "snmf_reftest", "nnls_24x18_k4", "nnls_reftest",
         "bnmf_96x64_k8", "bnmf_96x64_k8_f32", "bnmf_reftest", "bnmf_1024x256_k64"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    g = load_golden(name)
    cls = ORACLE[name.split("_")[0]]
    o = cls(g["V"], num_bases=int(g["k"]))
    o.W = g["W0"].copy()
    o.H = g["H0"].copy()
    o.factorize(niter=int(g["niter"]))
    f32 = g["W0"].dtype == np.float32
    # snmf_cfg5s_dense inverts an ill-conditioned 128x128 Gram in fp32 (k = n):
    # the reference's own fp32 digits are BLAS-order dependent there.
    tol = 5e-3 if name == "snmf_cfg5s_dense" else (2e-5 if f32 else 1e-9)
    if name.startswith("nnls"):
        tol = 1e-7       # scipy nnls vs exact active set on the Gram matrix
    assert len(o.ferr) == len(g["ferr"])
    assert rel_fro(o.W, g["W"]) < tol
    assert rel_fro(o.H, g["H"]) < tol
    np.testing.assert_allclose(o.ferr, g["ferr"], rtol=tol, atol=1e-9)
    assert o.W.dtype == g["W"].dtype and o.H.dtype == g["H"].dtype


@pytest.mark.parametrize("name", ["nnls_130x90_k33", "nnls_cfg3s", "nnls_300x200_k72", "nnls_260x300_k130"])
def test_nmfals_oracle_matches_the_real_nmfnnls_reference(name):
    """Round 4: pymf/nmfnnls.py -- the reference's own class, REAL scipy.optimize.nnls, no stand-in -- on
    the inputs of the nmfals_* / bigk_nmfals_* fixtures: identical objective (nmfnnls.py:69-80 vs nmfals.py:70-97), so
    these pin the NMFALS oracle's NUMBERS at cfg3's width and num_bases and beyond 64 / 128 bases; each must also
    agree with its stub-cvxopt sibling (which pins nmfals.py's data flow)."""
    g = load_golden(name)
    o = NMFALSOracle(g["V"], num_bases=int(g["k"]))
    o.W, o.H = g["W0"].copy(), g["H0"].copy()
    o.factorize(niter=int(g["niter"]))
    assert len(o.ferr) == len(g["ferr"])
    assert rel_fro(o.W, g["W"]) < 1e-7 and rel_fro(o.H, g["H"]) < 1e-7     # Lawson-Hanson on (A, b) vs exact active set on the Gram matrix
    np.testing.assert_allclose(o.ferr, g["ferr"], rtol=1e-7)
    sib = {"nnls_130x90_k33": "nmfals_130x90_k33", "nnls_cfg3s": "nmfals_cfg3s",
           "nnls_300x200_k72": "bigk_nmfals_300x200_k72", "nnls_260x300_k130": "bigk_nmfals_260x300_k130"}[name]
    h = load_golden(sib)
    assert np.array_equal(h["W0"], g["W0"]) and np.array_equal(h["V"], g["V"])
    assert rel_fro(h["W"], g["W"]) < 1e-7 and rel_fro(h["H"], g["H"]) < 1e-7


@pytest.mark.parametrize("name", ["nmfals_24x18_k4", "nmfals_reftest", "nmfals_130x90_k33", "nmfals_cfg3s"])
def test_nmfals_oracle_data_flow_pin_against_nmfals_py_with_stub_cvxopt(name):
    """Goldens from pymf/nmfals.py ITSELF (nmfals.py:70-97) driven through an exact-QP stand-in for
    cvxopt (gen_golden.load_reference_nmfals): pins the restatement's data flow -- -W.T / -H signs,
    float64 forcing, column scatter :75, row scatter :90 -- not cvxopt's interior-point digits."""
    g = load_golden(name)
    o = NMFALSOracle(g["V"], num_bases=int(g["k"]))
    o.W, o.H = g["W0"].copy(), g["H0"].copy()
    o.factorize(niter=int(g["niter"]))
    assert len(o.ferr) == len(g["ferr"])
    if name == "nmfals_reftest":
        # rank-3 data, 4 bases: singular Gram matrices, the minimisers are not unique -- what is
        # determined is the reconstruction and the error curve (the reference test's own bound, :86-88)
        assert rel_fro(o.W.dot(o.H), g["W"].dot(g["H"])) < 1e-5
        np.testing.assert_allclose(o.ferr, g["ferr"], rtol=1e-3, atol=1e-6)
        return
    assert rel_fro(o.W, g["W"]) < 1e-7 and rel_fro(o.H, g["H"]) < 1e-7
    np.testing.assert_allclose(o.ferr, g["ferr"], rtol=1e-7)
    if name == "nmfals_24x18_k4":          # same problem as the NNLS sibling's golden: same minimisers
        h = load_golden("nnls_24x18_k4")
        assert rel_fro(g["W"], h["W"]) < 1e-7 and rel_fro(g["H"], h["H"]) < 1e-7


@pytest.mark.parametrize("name", ["snmf_cfg5s_dense_f64", "snmf_csr_k128_f64"])
def test_snmf_oracle_at_cfg5_conditioning(name):
    """k = n = 128 (cfg5's shape class): cond(H H^T) ~ 1e7.  The float64 restatement reproduces the
    float64-default reference; the reference's OWN all-float32 run (kept in the fixture) is off by
    percents in W -- the yardstick for the device path's float32 arithmetic."""
    g = load_golden(name)
    o = SNMFOracle(g["V"], num_bases=int(g["k"]))
    o.W, o.H = g["W0"].copy(), g["H0"].copy()
    o.factorize(niter=int(g["niter"]))
    assert len(o.ferr) == len(g["ferr"]) == 2
    assert rel_fro(o.W, g["W"]) < 1e-6 and rel_fro(o.H, g["H"]) < 1e-6      # inv() digits are LAPACK-order dependent
    assert np.all(o.ferr < 1e-6 * np.linalg.norm(g["V"]))                   # the fit is exact (k = n)
    drift_w, drift_h = rel_fro(g["W32"], g["W"]), rel_fro(g["H32"], g["H"])
    assert drift_w > 1e-2 and drift_h > 5e-6, (drift_w, drift_h)            # the reference's own float32 path


@pytest.mark.parametrize("name", ["rnmf_60x40_k4", "rnmf_300x256_k32", "rnmf_300x256_k8"])
def test_rnmf_oracle_matches_reference_golden(name):
    """RNMF goldens come from the reference's own lazy init path (seed -> init_w -> init_h -> S)."""
    from oracle import RNMFOracle
    g = load_golden(name)
    np.random.seed(int(g["seed"]))
    o = RNMFOracle(g["V"], num_bases=int(g["k"]), lamb=float(g["lamb"]))
    o.factorize(niter=int(g["niter"]))
    assert rel_fro(o.W, g["W"]) < 1e-12 and rel_fro(o.H, g["H"]) < 1e-12 and rel_fro(o.S, g["S"]) < 1e-12
    np.testing.assert_allclose(o.ferr, g["ferr"], rtol=1e-12)


def test_known_answers_from_survey():
    """SURVEY.md section 8(c) 'known answers' measured on the reference."""
    g = load_golden("nmf_cfg1_f64")
    assert abs(g["ferr"][0] - 20.695357387) < 1e-6
    assert abs(g["ferr"][-1] - 18.846608158) < 1e-6
    assert abs(g["W"].sum() - 88.449976741) < 1e-5
    assert abs(g["H"].sum() - 113.099791227) < 1e-5
    g = load_golden("snmf_cfg1_f64")
    assert abs(g["ferr"][0] - 20.150047626) < 1e-6
    assert abs(g["ferr"][-1] - 18.870949944) < 1e-6


def test_reference_test_bound():
    """tests/test_pymf.py:86-88 -- ferr[-1]/(rows+cols) < 0.1 on the reference's data."""
    for name in ("nmf_reftest", "snmf_reftest", "nnls_reftest"):
        g = load_golden(name)
        assert g["ferr"][-1] / (g["V"].shape[0] + g["V"].shape[1]) < 0.1


def test_flag_sequence_and_resume():
    """tests/test_pymf.py:92-95 -- repeated factorize() calls with flags, W/H persist."""
    g = load_golden("nmf_flagseq")
    np.random.seed(int(g["seed"]))
    o = NMFOracle(g["V"], num_bases=int(g["k"]))
    o.factorize(niter=5)
    assert rel_fro(o.W, g["W_a"]) < 1e-9 and rel_fro(o.H, g["H_a"]) < 1e-9
    o.factorize(niter=5, compute_h=False)
    assert rel_fro(o.W, g["W_b"]) < 1e-9 and np.array_equal(o.H, g["H_a"]) or rel_fro(o.H, g["H_b"]) < 1e-9
    o.factorize(niter=5, compute_w=False)
    assert rel_fro(o.W, g["W_c"]) < 1e-9 and rel_fro(o.H, g["H_c"]) < 1e-9
    ferr_before = o.ferr.copy()
    o.factorize(niter=5, compute_err=False)
    assert rel_fro(o.W, g["W_d"]) < 1e-9 and rel_fro(o.H, g["H_d"]) < 1e-9
    np.testing.assert_array_equal(o.ferr, ferr_before)       # ferr untouched (nmf.py:179-180)
    np.testing.assert_allclose(o.ferr, g["ferr_d"], rtol=1e-9)


def test_early_exit_truncates_ferr():
    """nmf.py:198-202 -- converged at i: ferr = ferr[:i] (entry i dropped)."""
    g = load_golden("nmf_earlyexit")
    o = NMFOracle(g["V"], num_bases=2)
    o.W = g["W0"].copy()
    o.H = g["H0"].copy()
    o.factorize(niter=20, compute_w=False)
    assert len(o.ferr) == len(g["ferr"]) == 2
    np.testing.assert_allclose(o.H, g["H"], rtol=1e-9)


def test_sentinel_without_factors():
    o = NMFOracle(np.ones((3, 4)), num_bases=2)
    assert o.frobenius_norm() == -123456


NNDSVD_CASES = ["nndsvd_300x40_k6", "nndsvd_40x300_k6", "nndsvd_cfg1_k4", "nndsvd_37x29_k5", "nndsvd_doc_k2",
                "nndsvd_1024x256_k64"]


@pytest.mark.parametrize("name", NNDSVD_CASES)
def test_nndsvd_oracle_matches_reference_golden(name):
    """The restatement (with the reference's per-basis second SVD) and the closed form the device
    path evaluates both reproduce pymf.NNDSVD: the restatement exactly, the closed form (evaluated in
    float64) to the noise of the reference's float32 Gram matrix when the data are float32."""
    from oracle import NNDSVDOracle, nndsvd_closed_form
    g = load_golden(name)
    k = int(g["k"])
    o = NNDSVDOracle(g["V"], num_bases=k)
    o.factorize(niter=7, compute_w=False)                # arguments are overridden (nndsvd.py:111-114)
    assert len(o.ferr) == 1
    assert rel_fro(o.W, g["W"]) < 1e-12 and rel_fro(o.H, g["H"]) < 1e-12
    np.testing.assert_allclose(o.ferr, g["ferr"], rtol=1e-5)
    Wc, Hc = nndsvd_closed_form(g["V"], k)
    tolc = 1e-10 if g["V"].dtype == np.float64 else 2e-4
    assert rel_fro(Wc, g["W"]) < tolc and rel_fro(Hc, g["H"]) < tolc
    assert (Wc >= 0).all() and (Hc >= 0).all()


# ---- beyond the register-resident base counts (SNMF / RNMF / NNDSVD > 128, NMFALS > 64): `bigk_*` goldens ----
def test_oracles_match_reference_goldens_at_wide_base_counts():
    from oracle import RNMFOracle, NNDSVDOracle, nndsvd_closed_form
    g = load_golden("bigk_snmf_512x320_k160")
    o = SNMFOracle(g["V"], num_bases=int(g["k"]))
    o.W, o.H = g["W0"].copy(), g["H0"].copy()
    o.factorize(niter=int(g["niter"]))
    assert rel_fro(o.W, g["W"]) < 1e-8 and rel_fro(o.H, g["H"]) < 1e-8
    np.testing.assert_allclose(o.ferr, g["ferr"], rtol=1e-9)
    for name in ("bigk_nmfals_300x200_k72", "bigk_nmfals_260x300_k130"):
        g = load_golden(name)
        o = NMFALSOracle(g["V"], num_bases=int(g["k"]))
        o.W, o.H = g["W0"].copy(), g["H0"].copy()
        o.factorize(niter=int(g["niter"]))
        assert rel_fro(o.W, g["W"]) < 1e-6 and rel_fro(o.H, g["H"]) < 1e-6, name
        np.testing.assert_allclose(o.ferr, g["ferr"], rtol=1e-7)
    g = load_golden("bigk_rnmf_300x256_k140")
    np.random.seed(int(g["seed"]))
    o = RNMFOracle(g["V"], num_bases=int(g["k"]), lamb=float(g["lamb"]))
    o.factorize(niter=int(g["niter"]))
    assert rel_fro(o.W, g["W"]) < 1e-12 and rel_fro(o.H, g["H"]) < 1e-12 and rel_fro(o.S, g["S"]) < 1e-12
    g = load_golden("bigk_nndsvd_500x300_k150")
    o = NNDSVDOracle(g["V"], num_bases=150)
    o.factorize()
    assert rel_fro(o.W, g["W"]) < 1e-12 and rel_fro(o.H, g["H"]) < 1e-12
    Wc, Hc = nndsvd_closed_form(g["V"], 150)
    assert rel_fro(Wc, g["W"]) < 2e-3 and rel_fro(Hc, g["H"]) < 2e-3      # float32 Gram matrix in the reference, small gaps
