"""GPU parity for BNMF (SURVEY 8(f) 'next' row 1): reference goldens + oracle, through the C ABI."""
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


@pytest.mark.parametrize("name", ["bnmf_96x64_k8", "bnmf_96x64_k8_f32", "bnmf_reftest", "bnmf_1024x256_k64"])
def test_bnmf_vs_reference_golden(pm, name):
    g = load_golden(name)
    mdl = pm.BNMF(g["V"], num_bases=int(g["k"]))
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert len(mdl.ferr) == len(g["ferr"])
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 2e-6
    assert rel_fro(mdl.H, g["H"], what="mdl.H") < 2e-6
    close(mdl.ferr, g["ferr"], rtol=1e-5, what="mdl.ferr")
    # the lambda schedule: 1/niter * 1.1**niter on both (bnmf.py:84-85,118-119)
    expect = (1.0 / int(g["niter"])) * 1.1 ** int(g["niter"])
    assert abs(mdl._lamb_W - expect) < 1e-12 and abs(mdl._lamb_H - expect) < 1e-12
    if name == "bnmf_reftest":
        assert mdl.ferr[-1] / (g["V"].shape[0] + g["V"].shape[1]) < 0.1   # tests/test_pymf.py:86-88


def test_bnmf_hooks_and_flags_vs_oracle(pm):
    from oracle import BNMFOracle
    rs = np.random.RandomState(3)
    V = (rs.random_sample((300, 200)) < 0.25).astype(np.float32)
    W0, H0 = rs.random_sample((300, 24)), rs.random_sample((24, 200))
    mdl = pm.BNMF(V, num_bases=24)
    ref = BNMFOracle(V, num_bases=24)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    ref.W, ref.H = W0.copy(), H0.copy()
    with pytest.raises(AttributeError):
        mdl.update_w()                     # _lamb_W does not exist before factorize (bnmf.py:88)
    mdl.factorize(niter=4)
    ref.factorize(niter=4)
    mdl.factorize(niter=3, compute_h=False)          # lambdas do not move without update_h
    ref.factorize(niter=3, compute_h=False)
    assert abs(mdl._lamb_W - ref._lamb_W) < 1e-15
    mdl.update_w(); ref.update_w()
    mdl.update_h(); ref.update_h()
    assert abs(mdl._lamb_H - ref._lamb_H) < 1e-15
    assert rel_fro(mdl.W, ref.W, what="mdl.W") < 2e-6 and rel_fro(mdl.H, ref.H, what="mdl.H") < 6e-7
    assert abs(mdl.frobenius_norm() - ref.frobenius_norm()) / ref.frobenius_norm() < 2e-5


def test_bnmf_fused_and_tiled_agree(pm):
    from pymf_amd import _lib
    rs = np.random.RandomState(2)
    m, n, k = 8192, 256, 64
    V = (rs.random_sample((m, n)) < 0.2).astype(np.float32)
    W0 = rs.random_sample((m, k)).astype(np.float32)
    H0 = rs.random_sample((k, n)).astype(np.float32)
    a = _lib.Context(_lib.ALGO_BNMF, m, n, k)
    assert "bnmf" in a.path_name
    a.set_v_dense(V); a.set_w(W0); a.set_h(H0); a.set_lambda(0.1, 0.1)
    a.factorize(3, compute_err=False)
    b = _lib.Context(_lib.ALGO_BNMF, m, n, k)
    b.set_v_dense(V); b.set_w(W0); b.set_h(H0); b.set_lambda(0.1, 0.1)
    for _ in range(3):
        b.update_w()
        b.update_h()
    assert a.get_lambda() == b.get_lambda()
    assert rel_fro(a.get_w(), b.get_w(), what="a.get_w()") < 1e-9 and rel_fro(a.get_h(), b.get_h(), what="a.get_h()") < 1e-9


@pytest.mark.parametrize("shape,k", [((3000, 512), 32), ((2000, 384), 17), ((2500, 500), 9), ((4000, 190), 64),
                                     ((1500, 320), 30)])
def test_bnmf_every_fused_shape_class_vs_oracle(pm, shape, k):
    """BNMF on the shape classes of the fused kernel beyond the 4-panel one: 3/5/6 panels and the
    two-waves-per-block form (n up to 512 for k <= 32)."""
    from oracle import BNMFOracle
    from pymf_amd import _lib
    rs = np.random.RandomState(shape[1] + k)
    V = (rs.random_sample(shape) < 0.3).astype(np.float32)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    mdl = pm.BNMF(V, num_bases=k)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    mdl.factorize(niter=4)
    assert mdl._ctx.path_name.startswith("k_nmf_fused") and ",bnmf" in mdl._ctx.path_name
    o = BNMFOracle(V, num_bases=k)
    o.W, o.H = W0.copy(), H0.copy()
    o.factorize(niter=4)
    assert rel_fro(mdl.W, o.W, what="mdl.W") < 2e-6 and rel_fro(mdl.H, o.H, what="mdl.H") < 6e-7
    close(mdl.ferr, o.ferr, rtol=5e-9, what="mdl.ferr")


def test_bnmf_more_than_128_bases(pm):
    from oracle import BNMFOracle
    rs = np.random.RandomState(12)
    V = (rs.random_sample((1200, 260)) < 0.3).astype(np.float32)
    W0, H0 = rs.random_sample((1200, 200)), rs.random_sample((200, 260))
    mdl = pm.BNMF(V, num_bases=200)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    mdl.factorize(niter=4)
    o = BNMFOracle(V, num_bases=200)
    o.W, o.H = W0.copy(), H0.copy()
    o.factorize(niter=4)
    assert rel_fro(mdl.W, o.W, what="mdl.W") < 2e-6 and rel_fro(mdl.H, o.H, what="mdl.H") < 1e-6
    close(mdl.ferr, o.ferr, rtol=1e-9, what="mdl.ferr")
    assert abs(mdl._lamb_W - o._lamb_W) < 1e-12
