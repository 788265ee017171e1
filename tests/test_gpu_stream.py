"""GPU parity for the streamed (out-of-core) NMF path, SURVEY 8(f) row 4: `data` is read tile by
tile through `data[r0:r1, :]` and never made resident.  The arithmetic is the resident path's (same
W-step kernel per tile, W^T V | W^T W accumulated over the tiles in float64), so the tolerances of
tests/test_gpu_parity.py apply: 2e-5 relative Frobenius on W, H and 1e-5 on ferr."""
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


class SliceOnly(object):
    """An h5py-dataset-like source: has .shape, serves row slices, refuses to be read whole."""

    def __init__(self, arr, max_rows):
        self._a, self.shape, self.max_rows, self.reads = arr, arr.shape, max_rows, 0

    def __getitem__(self, key):
        rs = key[0] if isinstance(key, tuple) else key
        start, stop, _ = rs.indices(self.shape[0])
        assert stop - start <= self.max_rows, "the streamed path must never read more than a tile"
        self.reads += 1
        return self._a[key]


@pytest.mark.parametrize("name,rows", [("nmf_cfg4s", 512), ("nmf_cfg2s", 256), ("nmf_512x128_k16", 64),
                                       ("nmf_37x29_k5", 64), ("nmf_cfg1_f64", 64)])
def test_streamed_nmf_vs_reference_golden(pm, name, rows):
    g = load_golden(name)
    src = SliceOnly(g["V"], rows)
    mdl = pm.NMF(src, num_bases=int(g["k"]))
    mdl.stream_rows = rows
    mdl.W = g["W0"].copy()
    mdl.H = g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert src.reads >= int(g["niter"]) * ((g["V"].shape[0] + rows - 1) // rows)
    assert len(mdl.ferr) == len(g["ferr"])
    close(mdl.ferr, g["ferr"], rtol=5e-7, what="mdl.ferr")
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 4e-6 and rel_fro(mdl.H, g["H"], what="mdl.H") < 2e-6


@pytest.mark.parametrize("shape,k,rows", [((5000, 200), 20, 1024), ((1000, 70), 7, 384), ((4096, 256), 64, 4096),
                                          ((130, 300), 33, 64)])
def test_streamed_equals_resident(pm, shape, k, rows, tmp_path):
    rs = np.random.RandomState(shape[0] + k)
    V = rs.random_sample(shape).astype(np.float32)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    path = str(tmp_path / "v.f32")
    V.tofile(path)
    Vmm = np.memmap(path, dtype=np.float32, mode="r", shape=shape)       # out-of-core source
    res = pm.NMF(V, num_bases=k)
    res.W, res.H = W0.copy(), H0.copy()
    res.factorize(niter=6)
    st = pm.NMF(Vmm, num_bases=k)
    st.stream_rows = rows
    st.W, st.H = W0.copy(), H0.copy()
    st.factorize(niter=6)
    close(st.ferr, res.ferr, rtol=1e-7, what="st.ferr")
    assert rel_fro(st.W, res.W, what="st.W") < 6e-7 and rel_fro(st.H, res.H, what="st.H") < 5e-7
    assert abs(st.frobenius_norm() - res.frobenius_norm()) <= 2e-6 * res.frobenius_norm()


@pytest.mark.parametrize("cls_name,shape,k,rows", [("NMF", (3000, 256), 200, 1024), ("NMF", (2000, 192), 130, 512),
                                                  ("BNMF", (2048, 128), 140, 640), ("SNMF", (3000, 320), 150, 1024),
                                                  ("NMFALS", (700, 160), 130, 256), ("NMFALS", (900, 200), 100, 256)])
def test_streamed_equals_resident_beyond_128_bases(pm, cls_name, shape, k, rows, tmp_path):
    """num_bases > 128 (the products in blocks of 128 bases; NMFALS also at 100 bases: k_nnqp_wave per tile)
    through the row-tile passes: the same
    iteration as on resident data (per tile the W rule, W_b^T V and W_b^T W partials accumulated in float64)."""
    rs = np.random.RandomState(shape[0] + k)
    V = rs.random_sample(shape).astype(np.float32)
    if cls_name == "BNMF":
        V = (V < 0.3).astype(np.float32)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    path = str(tmp_path / "v.f32")
    V.tofile(path)
    Vmm = np.memmap(path, dtype=np.float32, mode="r", shape=shape)
    cls = getattr(pm, cls_name)
    niter = 3 if cls_name == "NMFALS" else 5
    res = cls(V, num_bases=k)
    res.W, res.H = W0.copy(), H0.copy()
    res.factorize(niter=niter)
    st = cls(Vmm, num_bases=k)
    st.stream_rows = rows
    st.W, st.H = W0.copy(), H0.copy()
    st.factorize(niter=niter)
    assert st._ctx.path_name and len(st.ferr) == len(res.ferr)
    tol = 2e-5 if cls_name in ("SNMF", "NMFALS") else 2e-6
    close(st.ferr, res.ferr, rtol=tol, what="st.ferr")
    assert rel_fro(st.W, res.W, what="st.W") < tol and rel_fro(st.H, res.H, what="st.H") < tol


def test_streamed_flags_and_hooks(pm):
    from oracle import NMFOracle
    rs = np.random.RandomState(9)
    V = rs.random_sample((700, 90)).astype(np.float32)
    W0, H0 = rs.random_sample((700, 9)), rs.random_sample((9, 90))
    for flags in (dict(compute_w=False), dict(compute_h=False), dict(compute_err=False)):
        o = NMFOracle(V, num_bases=9)
        o.W, o.H = W0.copy(), H0.copy()
        o.factorize(niter=4, **flags)
        m = pm.NMF(V, num_bases=9)
        m.stream_rows = 256
        m.W, m.H = W0.copy(), H0.copy()
        m.factorize(niter=4, **flags)
        assert rel_fro(m.W, o.W, what="m.W") < 2e-6 and rel_fro(m.H, o.H, what="m.H") < 4e-7, flags
        if flags.get("compute_err", True):
            close(m.ferr, o.ferr, rtol=6e-8, what="m.ferr")
        else:
            assert not hasattr(m, "ferr")
    o = NMFOracle(V, num_bases=9)
    o.W, o.H = W0.copy(), H0.copy()
    m = pm.NMF(V, num_bases=9)
    m.stream_rows = 128
    m.W, m.H = W0.copy(), H0.copy()
    for _ in range(2):
        m.update_w(); o.update_w()
        m.update_h(); o.update_h()
    assert rel_fro(m.W, o.W, what="m.W") < 7e-7 and rel_fro(m.H, o.H, what="m.H") < 3e-7
    assert abs(m.frobenius_norm() - o.frobenius_norm()) <= 1e-5 * o.frobenius_norm()


def test_streamed_early_exit_on_exact_data(pm):
    """nmf.py:198-202 on exact data: the trace identity cancels, the direct residual pass takes over
    and the loop stops with len(ferr) == 2 exactly as the reference does."""
    g = load_golden("nmf_earlyexit")
    mdl = pm.NMF(g["V"], num_bases=int(g["k"]))
    mdl.stream_rows = 64
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]), compute_w=False)
    assert len(mdl.ferr) == len(g["ferr"]) == 2
    assert rel_fro(mdl.H, g["H"], what="mdl.H") < 3e-9


def test_stream_c_abi_contract(pm):
    from pymf_amd import _lib
    from pymf_amd._lib import PmfError
    V = np.random.RandomState(1).random_sample((300, 40)).astype(np.float32)
    ctx = _lib.Context(_lib.ALGO_NMF, 300, 40, 5)
    with pytest.raises(PmfError):
        ctx.stream_begin()                               # W, H not set
    ctx.set_w(np.random.RandomState(2).random_sample((300, 5)))
    ctx.set_h(np.random.RandomState(3).random_sample((5, 40)))
    with pytest.raises(PmfError):
        ctx.stream_tile(0, V[:64])                       # no pass open
    ctx.stream_begin(max_tile_rows=128)
    with pytest.raises(PmfError):
        ctx.stream_tile(64, V[64:128])                   # out of order
    with pytest.raises(PmfError):
        ctx.stream_tile(0, V[:100])                      # inner tile not a multiple of 64 rows
    with pytest.raises(PmfError):
        ctx.stream_tile(0, V[:192])                      # larger than max_tile_rows
    ctx.stream_tile(0, V[:128])
    with pytest.raises(PmfError):
        ctx.stream_end()                                 # rows missing
    with pytest.raises(PmfError):
        ctx.update_w()                                   # V was never made resident
    rn = _lib.Context(_lib.ALGO_RNMF, 300, 40, 5)
    rn.set_w(np.ones((300, 5))); rn.set_h(np.ones((5, 40)))
    with pytest.raises(PmfError):
        rn.stream_begin()                                # RNMF keeps an m x n state (S): not streamed
    ctx.close(); rn.close()


@pytest.mark.parametrize("name,rows", [("bnmf_96x64_k8", 64), ("bnmf_1024x256_k64", 256)])
def test_streamed_bnmf_vs_reference_golden(pm, name, rows):
    """BNMF streams too (bnmf.py:79-90 reads data[:, :] like NMF): same tiles, the penalised epilogue, the
    reference's lambda schedule (both weights grow by 1.1 per update_h)."""
    g = load_golden(name)
    src = SliceOnly(g["V"], rows)
    mdl = pm.BNMF(src, num_bases=int(g["k"]))
    mdl.stream_rows = rows
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert src.reads >= int(g["niter"]) * ((g["V"].shape[0] + rows - 1) // rows)
    assert len(mdl.ferr) == len(g["ferr"])
    close(mdl.ferr, g["ferr"], rtol=3e-7, what="mdl.ferr")
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < 2e-6 and rel_fro(mdl.H, g["H"], what="mdl.H") < 2e-6
    res = pm.BNMF(g["V"], num_bases=int(g["k"]))
    res.W, res.H = g["W0"].copy(), g["H0"].copy()
    res.factorize(niter=int(g["niter"]))
    assert (mdl._lamb_W, mdl._lamb_H) == (res._lamb_W, res._lamb_H)


@pytest.mark.parametrize("cls_name,name,rows", [("SNMF", "snmf_cfg4s", 512), ("SNMF", "snmf_512x128_k16", 128),
                                                ("SNMF", "snmf_37x29_k5", 64), ("NMFALS", "nmfals_130x90_k33", 64),
                                                ("NMFALS", "nmfals_cfg3s", 512), ("NMFNNLS", "nnls_24x18_k4", 64)])
def test_streamed_snmf_and_nmfals_vs_reference_golden(pm, cls_name, name, rows):
    """The `data[:, :]` idiom is in every class of the reference (snmf.py:68,79; nmfals.py:73,88): SNMF and
    NMFALS stream row tiles like NMF -- per tile the class's W step (W = V M^T resp. one QP per row), then
    the partials of W^T V | W^T W; the H step once per pass."""
    g = load_golden(name)
    src = SliceOnly(g["V"], rows)
    mdl = getattr(pm, cls_name)(src, num_bases=int(g["k"]))
    mdl.stream_rows = rows
    mdl.W, mdl.H = g["W0"].copy(), g["H0"].copy()
    mdl.factorize(niter=int(g["niter"]))
    assert src.reads >= int(g["niter"]) * ((g["V"].shape[0] + rows - 1) // rows)
    assert len(mdl.ferr) == len(g["ferr"])
    tol = 5e-5 if cls_name == "SNMF" else 2e-4
    assert rel_fro(mdl.W, g["W"], what="mdl.W") < tol and rel_fro(mdl.H, g["H"], what="mdl.H") < tol
    close(mdl.ferr, g["ferr"], rtol=5e-7, what="mdl.ferr")
    # and the streamed object equals the resident one on the same inputs
    res = getattr(pm, cls_name)(g["V"], num_bases=int(g["k"]))
    res.W, res.H = g["W0"].copy(), g["H0"].copy()
    res.factorize(niter=int(g["niter"]))
    assert rel_fro(mdl.W, res.W, what="streamed vs resident W") < tol
    assert rel_fro(mdl.H, res.H, what="streamed vs resident H") < tol
    assert abs(mdl.frobenius_norm() - res.frobenius_norm()) <= 1e-4 * res.frobenius_norm()


# ---- host <-> device transport (round 4): float64 arrays cross as they are and are rounded / widened on the device -------
def test_float64_transport_equals_host_side_conversion():
    """pmf_set_w_f64 / pmf_set_h_f64 / pmf_set_v_dense_f64 round on the device what np.astype(float32) rounds on the host
    (round to nearest even both): the device copies must be bit-identical; pmf_get_*_f64 widens exactly; a pitched host
    array (leading dimension > n) and ragged shapes (padding columns / rows) go through the same paths."""
    import ctypes
    from pymf_amd import _lib
    rs = np.random.RandomState(3)
    for (m, n, k) in ((1000, 200, 33), (4096, 256, 64), (77, 50, 7)):
        V64 = rs.random_sample((m, n)) - 0.25
        W64 = rs.random_sample((m, k)) * 3.0
        H64 = rs.random_sample((k, n)) + 1e-3
        a = _lib.Context(_lib.ALGO_SNMF, m, n, k)        # SNMF: mixed-sign data allowed
        b = _lib.Context(_lib.ALGO_SNMF, m, n, k)
        a.set_v_dense(V64); a.set_w(W64); a.set_h(H64)                                   # float64 paths
        b.set_v_dense(V64.astype(np.float32)); b.set_w(W64.astype(np.float32)); b.set_h(H64.astype(np.float32))
        np.testing.assert_array_equal(a.get_w(), b.get_w())
        np.testing.assert_array_equal(a.get_h(), b.get_h())
        np.testing.assert_array_equal(a.get_w(), W64.astype(np.float32))
        assert a.frobenius() == b.frobenius()                                            # same V, W, H on the device
        # widening on the device == widening on the host; written into the caller's array in place
        Wout = np.full((m, k), np.nan)
        Hout = np.full((k, n), np.nan)
        assert a.get_w_into(Wout) and a.get_h_into(Hout)
        np.testing.assert_array_equal(Wout, W64.astype(np.float32).astype(np.float64))
        # SNMF keeps H in FLOAT64 on the device (round 6; the reference's H is float64, nmf.py:120, and inv(H H^T) amplifies its
        # rounding): a float64 H comes back as it went up, bit for bit; its float32 view is the rounding
        np.testing.assert_array_equal(Hout, H64)
        np.testing.assert_array_equal(a.get_h(), H64.astype(np.float32))
        a.set_h(H64.astype(np.float32))                                                  # a float32 upload replaces it: widened
        assert a.get_h_into(Hout)
        np.testing.assert_array_equal(Hout, H64.astype(np.float32).astype(np.float64))
        a.set_h(H64)
        cn = _lib.Context(_lib.ALGO_NMF, m, n, k)                                        # every other class: H is float32 on the device
        cn.set_h(H64)
        assert cn.get_h_into(Hout)
        np.testing.assert_array_equal(Hout, H64.astype(np.float32).astype(np.float64))
        cn.close()
        W32o = np.empty((m, k), dtype=np.float32)
        assert a.get_w_into(W32o) and np.array_equal(W32o, b.get_w())
        assert not a.get_w_into(np.empty((m, k + 1)))                                    # wrong shape: the caller copies
        assert not a.get_w_into(np.empty((m, 2 * k))[:, ::2])                            # not contiguous
        # a pitched float32 host array through the C ABI itself (ld > n)
        wide = np.zeros((m, n + 13), dtype=np.float32)
        wide[:, :n] = V64.astype(np.float32)
        lib = _lib.load()
        assert lib.pmf_set_v_dense_f32(b._h, wide.ctypes.data, n + 13) == 0
        assert a.frobenius() == b.frobenius()
        # ... and the float64 one
        wide64 = np.zeros((m, n + 5))
        wide64[:, :n] = V64
        assert lib.pmf_set_v_dense_f64(b._h, wide64.ctypes.data, n + 5) == 0
        assert a.frobenius() == b.frobenius()
        a.close(); b.close()


def test_first_call_timings_are_reported():
    import pymf_amd
    rs = np.random.RandomState(4)
    V = rs.random_sample((2048, 256)).astype(np.float32)
    mdl = pymf_amd.NMF(V, num_bases=16)
    mdl.W, mdl.H = rs.random_sample((2048, 16)), rs.random_sample((16, 256))
    mdl.factorize(niter=3, compute_err=False)
    t = mdl.last_call_ms
    assert set(t) >= {"ctx", "init", "upload", "loop", "total"} and t["total"] >= t["loop"] > 0.0 and t["upload"] > 0.0
    mdl.factorize(niter=3, compute_err=False)
    assert "ctx" not in mdl.last_call_ms or mdl.last_call_ms["ctx"] == 0.0            # the context exists already


@pytest.mark.parametrize("cls_name,shape,k,rows", [("NMF", (600, 100), 130, 256), ("NMF", (600, 1100), 130, 256), ("SNMF", (600, 256), 200, 64),
                                                    ("BNMF", (600, 256), 130, 256)])
def test_streamed_frobenius_norm_beyond_128_bases(pm, cls_name, shape, k, rows):
    """Round 4 (found by tests/sweeps/fuzz_sequences.py): frobenius_norm() of a streamed object -- the residual-only pass over
    the row tiles, which a subclass that overrides a hook also runs once per iteration -- used the <= 128-base residual kernel
    at any base count: beyond 128 bases it read W with the wrong row stride and stopped at 128 bases (5 621 for 7 828)."""
    import oracle
    from pymf_amd.bnmf import BNMF
    rs = np.random.RandomState(sum(shape) + k)
    V = rs.random_sample(shape).astype(np.float32) - (0.4 if cls_name == "SNMF" else 0.0)
    W0, H0 = rs.random_sample((shape[0], k)), rs.random_sample((k, shape[1]))
    cls = BNMF if cls_name == "BNMF" else getattr(pm, cls_name)
    a = cls(V.copy(), num_bases=k)
    a.stream_rows = rows
    a.W, a.H = W0.copy(), H0.copy()
    f64 = lambda W, H: float(np.linalg.norm(V.astype(np.float64) - np.asarray(W, dtype=np.float64).dot(np.asarray(H, dtype=np.float64))))
    close(a.frobenius_norm(), f64(W0, H0), rtol=2e-6, what="streamed frobenius_norm at the start")
    if cls_name == "BNMF":
        a.factorize(niter=1)
    else:
        a.update_w(); a.update_h()
    close(a.frobenius_norm(), f64(a.W, a.H), rtol=2e-5, what="streamed frobenius_norm after one iteration")


@pytest.mark.parametrize("cls_name", ["NMF", "NMFALS", "SNMF"])
@pytest.mark.parametrize("layout", ["C float64", "F float32", "strided float32"])
def test_streamed_tiles_converted_on_the_fly(pm, cls_name, layout):
    """Round 4 (found by tests/sweeps/fuzz_sequences.py): `data` that is float64, Fortran-ordered or a strided view reaches
    pmf_stream_tile as a TEMPORARY float32 copy per tile.  The tile's host-to-device copy is enqueued behind the consumer of the
    device buffer's previous tile and read the host memory when it ran -- with the host many tiles ahead, long after the binding's
    three-deep keep-alive had let the temporary go: NaN / 1e17 factors (float32 C-ordered data, a view of the caller's array, was
    never affected).  pmf_stream_tile now waits for the copy of the tile two calls back, which makes the documented lifetime
    (valid until the next but one call) sufficient."""
    import oracle
    m, n, k = 3000, 300, 4
    rs = np.random.RandomState(1)
    Vn = rs.random_sample((m, n)).astype(np.float32) - (0.4 if cls_name == "SNMF" else 0.0)
    Vd = {"C float64": Vn.astype(np.float64), "F float32": np.asfortranarray(Vn), "strided float32": np.repeat(Vn, 2, axis=1)[:, ::2]}[layout]
    W0, H0 = rs.random_sample((m, k)), rs.random_sample((k, n))
    a, o = getattr(pm, cls_name)(Vd, num_bases=k), getattr(oracle, cls_name + "Oracle")(Vn.astype(np.float64), num_bases=k)
    a.stream_rows = 64
    a.W, a.H = W0.copy(), H0.copy(); o.W, o.H = W0.copy(), H0.copy()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a.factorize(niter=3)
    o.factorize(niter=3)
    tol = 2e-5 if cls_name != "NMFALS" else 1e-4
    assert rel_fro(a.W, o.W, what="mdl.W") < tol and rel_fro(a.H, o.H, what="mdl.H") < tol
    close(a.ferr, o.ferr, rtol=1e-5, what="mdl.ferr")


def test_error_only_streamed_pass_after_a_new_w(pm):
    """Round 4 (tests/sweeps/fuzz_abi_sequences.py): pmf_set_w_* dropped (P | S) but left the trace terms <P,H>, <S,G> of the last
    H step marked current; an error-only streamed pass then rebuilt (P | S) for the NEW W and evaluated ||V - W H|| from the OLD
    terms (79.31 for 79.60).  Directly on the C ABI: full streamed pass, new W, error-only streamed pass."""
    from pymf_amd import _lib
    from oracle import NMFOracle
    rs = np.random.RandomState(3)
    m, n, k = 700, 1100, 33
    V = rs.random_sample((m, n)).astype(np.float32)
    o = NMFOracle(V.astype(np.float64), num_bases=k)
    o.W, o.H = rs.random_sample((m, k)), rs.random_sample((k, n))
    c = _lib.Context(_lib.ALGO_NMF, m, n, k)
    c.set_w(o.W.copy()); c.set_h(o.H.copy())

    def streamed_pass(rows, **kw):
        c.stream_begin(max_tile_rows=rows, **kw)
        for r0 in range(0, m, rows):
            c.stream_tile(r0, V[r0:r0 + rows])
        return c.stream_end()[0]

    f1 = streamed_pass(256, compute_w=True, compute_h=True, compute_err=True)
    o.factorize(niter=1)
    close(f1, o.ferr[-1], rtol=1e-5, what="ferr of a full streamed pass")
    Wn = o.W * (1.0 + 0.1 * rs.random_sample(o.W.shape))
    c.set_w(Wn.copy()); o.W = Wn.copy()
    f2 = streamed_pass(64, compute_w=False, compute_h=False, compute_err=True)
    close(f2, o.frobenius_norm(), rtol=1e-5, what="error-only streamed pass after pmf_set_w")
    c.close()
