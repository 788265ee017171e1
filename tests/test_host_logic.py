"""Host-side logic of the drop-in classes that needs no GPU: which factorize() path is taken (template
method, nmf.py:182-202), the W / H attribute semantics, the digest-based change detection and the
streamed-mode switches."""
import numpy as np
import pytest

import pymf_amd
from pymf_amd import _lib
from pymf_amd.rnmf import RNMF


def test_shipped_classes_take_the_one_call_path():
    V = np.ones((6, 5), dtype=np.float32)
    for cls in (pymf_amd.NMF, pymf_amd.SNMF, pymf_amd.NMFALS, pymf_amd.NMFNNLS, pymf_amd.BNMF, RNMF):
        assert not cls(V, num_bases=2)._hooks_overridden(), cls.__name__


def test_overridden_hooks_are_detected():
    V = np.ones((6, 5), dtype=np.float32)

    class A(pymf_amd.NMF):
        def update_h(self):
            pymf_amd.NMF.update_h(self)

    class B(pymf_amd.SNMF):
        def converged(self, i):
            return False

    class C(pymf_amd.BNMF):          # a subclass that overrides nothing: still the fast path
        extra = 1

    class D(A):                      # inherits A's override
        pass

    assert A(V)._hooks_overridden() and B(V)._hooks_overridden() and D(V)._hooks_overridden()
    assert not C(V)._hooks_overridden()
    m = pymf_amd.NMF(V)
    m.frobenius_norm = lambda: 0.0   # instance-level replacement
    assert m._hooks_overridden()


def test_w_h_attribute_semantics():
    """W and H behave like the reference's plain attributes: absent until created (nmf.py:173-177 tests
    hasattr), assignable, deletable; assigning marks the device copy stale."""
    m = pymf_amd.NMF(np.ones((6, 5), dtype=np.float32), num_bases=2)
    assert not hasattr(m, "W") and not hasattr(m, "H")
    with pytest.raises(AttributeError):
        m.W
    W = np.random.rand(6, 2)
    m.W = W
    assert m.W is W and hasattr(m, "W") and m._w_fp is None
    m._w_fp = ("something",)
    m.W = W.copy()
    assert m._w_fp is None
    del m.W
    assert not hasattr(m, "W")
    with pytest.raises(AttributeError):
        del m.W
    assert m.frobenius_norm() == -123456          # nmf.py:112: no W/H yet -> the sentinel, no device needed


def test_fingerprint_follows_in_place_edits():
    from pymf_amd.nmf import _fingerprint
    W = np.random.RandomState(0).rand(1000, 8)
    f0 = _fingerprint(W)
    assert _fingerprint(W) == f0
    W[[1, 2]] = W[[2, 1]]
    assert _fingerprint(W) != f0


def test_stream_switches():
    V = np.ones((200, 5), dtype=np.float32)
    for cls, ok in ((pymf_amd.NMF, True), (pymf_amd.BNMF, True), (pymf_amd.SNMF, True), (pymf_amd.NMFALS, True), (RNMF, False)):
        m = cls(V, num_bases=2)
        assert m._stream_rows() == 0
        m.stream_rows = 100
        assert m._stream_rows() == (128 if ok else 0), cls.__name__       # rounded up to 64 rows


def test_row_span_single_rank():
    m = pymf_amd.NMF(np.ones((37, 5), dtype=np.float32), num_bases=2)
    assert m._row_span() == (0, 37, 37) and m._global_rows() == 37


class _CountingCtx(object):
    """Test double of _lib.Context (NumPy float64 math through the oracle's functions) that counts what
    crosses the boundary: uploads of W / H and downloads of them."""

    def __init__(self, m, n, k):
        self.m, self.n, self.k = m, n, k
        self.up = {"W": 0, "H": 0, "V": 0}
        self.down = {"W": 0, "H": 0}

    def set_v_dense(self, V):
        self.V = np.asarray(V, dtype=np.float64)
        self.up["V"] += 1

    def set_w(self, W):
        self.W = np.array(W, dtype=np.float64)
        self.up["W"] += 1

    def set_h(self, H):
        self.H = np.array(H, dtype=np.float64)
        self.up["H"] += 1

    def get_w(self):
        self.down["W"] += 1
        return self.W.astype(np.float32)

    def get_h(self):
        self.down["H"] += 1
        return self.H.astype(np.float32)

    def invalidate_v(self):
        pass

    def factorize(self, niter, compute_w=True, compute_h=True, compute_err=True, conv_eps=1e-8):
        import oracle
        for _ in range(niter):
            if compute_w:
                oracle.nmf_update_w(self.V, self.W, self.H)
            if compute_h:
                oracle.nmf_update_h(self.V, self.W, self.H)
        return (np.ones(max(niter, 1)) if compute_err else None), niter, -1


def _model_with_double(m=40, n=12, k=3, seed=0):
    rs = np.random.RandomState(seed)
    mdl = pymf_amd.NMF(rs.rand(m, n).astype(np.float32), num_bases=k)
    mdl._ctx = _CountingCtx(m, n, k)
    mdl.W, mdl.H = rs.rand(m, k), rs.rand(k, n)
    return mdl


def test_factors_stay_on_the_device_between_calls_when_nobody_can_see_the_host_arrays():
    """`mdl.factorize(); mdl.factorize(); mdl.W`: the host arrays belong to the object alone, so they are
    brought up to date when they are read -- one download of W and H in all, one upload."""
    import oracle
    mdl = _model_with_double()
    ref = oracle.NMFOracle(mdl.data, num_bases=3)
    ref.W, ref.H = mdl.__dict__["_W"].copy(), mdl.__dict__["_H"].copy()
    ctx = mdl._ctx
    mdl.factorize(niter=2, compute_err=False)
    mdl.factorize(niter=3, compute_err=False)
    assert ctx.up == {"W": 1, "H": 1, "V": 1} and ctx.down == {"W": 0, "H": 0}
    ref.factorize(niter=5, compute_err=False)
    np.testing.assert_allclose(mdl.W, ref.W, rtol=1e-6)      # the read refreshes (float32 transport)
    np.testing.assert_allclose(mdl.H, ref.H, rtol=1e-6)
    assert ctx.down == {"W": 1, "H": 1}
    mdl.factorize(niter=1, compute_err=False)                 # handed out, unchanged: digested, not uploaded
    assert ctx.up["W"] == 1 and ctx.up["H"] == 1


def test_held_arrays_are_updated_in_place_after_every_call():
    """`w = mdl.W; mdl.factorize(); w` has changed (nmf.py:131-132 writes in place): an array somebody
    else holds -- a name, or a view of it -- is refreshed eagerly."""
    mdl = _model_with_double(seed=1)
    w = mdl.W
    before = w.copy()
    mdl.factorize(niter=1, compute_err=False)
    assert mdl._ctx.down["W"] == 1 and not np.array_equal(w, before) and mdl.W is w
    assert mdl._ctx.down["H"] == 0                            # nobody holds H
    hv = mdl.H[:1]                                            # a live VIEW keeps the base reachable
    assert mdl._ctx.down["H"] == 1
    mdl.factorize(niter=1, compute_err=False)
    assert mdl._ctx.down["H"] == 2
    del hv


def test_in_place_edit_through_the_attribute_reaches_the_device():
    mdl = _model_with_double(seed=2)
    mdl.factorize(niter=1, compute_err=False)
    mdl.H[0, 0] += 1.0                                        # read (refresh) + edit in place
    expect = mdl.__dict__["_H"].copy()
    mdl.factorize(niter=0, compute_err=False)
    assert mdl._ctx.up["H"] == 2
    np.testing.assert_allclose(mdl._ctx.H, expect.astype(np.float32), rtol=1e-6)


def test_eager_factors_switch_and_views_of_user_memory():
    mdl = _model_with_double(seed=3)
    mdl.eager_factors = True
    mdl.factorize(niter=1, compute_err=False)
    assert mdl._ctx.down == {"W": 1, "H": 1}
    big = np.random.rand(40, 8)
    m2 = _model_with_double(seed=4)
    m2.W = big[:, :3]                                         # a view of the caller's buffer: never lazy
    assert m2._held_elsewhere("W")
    m2.factorize(niter=1, compute_err=False)
    assert m2._ctx.down["W"] == 1
    np.testing.assert_allclose(big[:, :3], m2._ctx.W, rtol=1e-6)


def test_pickling_flushes_the_host_arrays():
    import pickle
    mdl = _model_with_double(seed=5)
    mdl.factorize(niter=2, compute_err=False)
    assert mdl._host_stale == {"W", "H"}
    clone = pickle.loads(pickle.dumps(mdl))
    assert clone._ctx is None and clone._host_stale == set()
    np.testing.assert_allclose(clone.__dict__["_W"], mdl._ctx.W, rtol=1e-6)


def test_failed_w_step_leaves_the_previous_factors_like_the_reference():
    """snmf.py:69-70: np.linalg.inv raises BEFORE W is rebound, so after a LinAlgError at iteration i the object
    holds iteration i-1's W and H.  Here the failing call leaves garbage in the device W: before every W step
    that may fail the device W is snapshot (device to device) and put back when the step raised; the factors,
    still device-resident, reach the host when they are read."""
    import oracle

    class Ctx(_CountingCtx):
        calls = 0

        def snapshot_w(self):                           # pmf_snapshot_w: a device-to-device copy
            self.Wsnap = self.W.copy()

        def restore_w(self):
            self.W = self.Wsnap.copy()

        def update_w(self):
            Ctx.calls += 1
            if Ctx.calls == 3:
                self.W[:] = np.nan                      # what a singular H H^T leaves behind on the device
                raise _lib.PmfError("SNMF: H H^T is singular")
            oracle.nmf_update_w(self.V, self.W, self.H)

        def update_h(self):
            oracle.nmf_update_h(self.V, self.W, self.H)

    rs = np.random.RandomState(7)
    V = rs.rand(30, 10).astype(np.float32)
    mdl = pymf_amd.SNMF(V, num_bases=3)
    mdl._ctx = Ctx(30, 10, 3)
    W0, H0 = rs.rand(30, 3), rs.rand(3, 10)
    mdl.W, mdl.H = W0.copy(), H0.copy()
    with pytest.raises(_lib.PmfError):
        mdl.factorize(niter=5, show_progress=True, compute_err=False)     # hook loop; the third W step fails
    Wr, Hr = W0.copy(), H0.copy()
    for _ in range(2):
        oracle.nmf_update_w(V.astype(np.float64), Wr, Hr)
        oracle.nmf_update_h(V.astype(np.float64), Wr, Hr)
    np.testing.assert_allclose(mdl.W, Wr, rtol=1e-6)
    np.testing.assert_allclose(mdl.H, Hr, rtol=1e-6)
    assert np.isfinite(mdl.W).all()


def test_success_then_failing_one_call_factorize_keeps_a_consistent_pair():
    """Advisor (round 3): after a successful lazy factorize() W and H live on the device alone; a factorize() that
    raises afterwards must not declare the untouched (older) host arrays current.  SNMF (a class whose W step may
    fail): the pair before the failing call survives -- W through the device snapshot, H through the flush that
    precedes such a call.  NMF: factors that lived on the device alone keep doing so."""
    import oracle

    class Ctx(_CountingCtx):
        fail_next = False

        def snapshot_w(self):
            self.Wsnap = self.W.copy()

        def restore_w(self):
            self.W = self.Wsnap.copy()

        def factorize(self, niter, *a, **kw):
            if self.fail_next:
                self.W[:] = np.nan                      # what a singular H H^T leaves behind on the device
                self.H[:] = np.nan
                raise np.linalg.LinAlgError("Singular matrix")
            return _CountingCtx.factorize(self, niter, *a, **kw)

    rs = np.random.RandomState(11)
    V = rs.rand(30, 10).astype(np.float32)
    for cls in (pymf_amd.SNMF, pymf_amd.NMF):
        mdl = cls(V, num_bases=3)
        mdl._ctx = Ctx(30, 10, 3)
        W0, H0 = rs.rand(30, 3), rs.rand(3, 10)
        mdl.W, mdl.H = W0.copy(), H0.copy()
        mdl.factorize(niter=2, compute_err=False)                     # success: both factors device-only now
        assert mdl._host_stale == {"W", "H"}
        Wr, Hr = W0.copy(), H0.copy()
        for _ in range(2):
            oracle.nmf_update_w(V.astype(np.float64), Wr, Hr)
            oracle.nmf_update_h(V.astype(np.float64), Wr, Hr)
        mdl._ctx.fail_next = True
        with pytest.raises(np.linalg.LinAlgError):
            mdl.factorize(niter=3, compute_err=False)
        if cls is pymf_amd.SNMF:
            np.testing.assert_allclose(mdl.W, Wr, rtol=1e-6)          # call 1's result, not W0
            np.testing.assert_allclose(mdl.H, Hr, rtol=1e-6)          # ... and not H0
            # and the next call computes from that pair
            mdl._ctx.fail_next = False
            mdl.factorize(niter=0, compute_err=False)
            np.testing.assert_allclose(mdl._ctx.H, Hr.astype(np.float32), rtol=1e-6)
        else:
            # nothing better than the device copies exists: they stay the current ones (never W0 / H0 again)
            assert mdl._host_stale == {"W", "H"}
            assert not np.array_equal(mdl.__dict__["_W"], Wr)         # host array untouched, and not declared current
            assert mdl._w_fp is not None or "W" in mdl._host_stale


def test_copies_share_no_bookkeeping_with_the_original():
    import copy
    mdl = _model_with_double(seed=6)
    mdl.factorize(niter=1, compute_err=False)
    clone = copy.copy(mdl)
    assert clone._host_stale is not mdl._host_stale and clone._handed is not mdl._handed
    mdl.factorize(niter=1, compute_err=False)                         # stale again on the original only
    assert clone._host_stale == set() and clone._ctx is None


def test_float64_data_warns_once_per_object():
    import warnings
    mdl = _model_with_double(seed=8)
    mdl.data = mdl.data.astype(np.float64)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        mdl.factorize(niter=1, compute_err=False)
        mdl.check_data = True
        mdl.factorize(niter=1, compute_err=False)
    hits = [r for r in rec if issubclass(r.category, pymf_amd.nmf.PrecisionWarning)]
    assert len(hits) == 1 and "float32" in str(hits[0].message)


class _LateCtx(_CountingCtx):
    """The counting double with the four calls the late data check needs; a log of the boundary traffic in order."""

    def __init__(self, m, n, k):
        _CountingCtx.__init__(self, m, n, k)
        self.log, self.aborted = [], False

    def snapshot_w(self):
        self.log.append("snapshot_w"); self._ws = self.W.copy()

    def snapshot_h(self):
        self.log.append("snapshot_h"); self._hs = self.H.copy()

    def restore_w(self):
        self.log.append("restore_w"); self.W = self._ws.copy()

    def restore_h(self):
        self.log.append("restore_h"); self.H = self._hs.copy()

    def abort(self, on=True):
        self.log.append("abort(%d)" % (1 if on else 0)); self.aborted = bool(on)

    def set_v_dense(self, V):
        self.log.append("set_v"); _CountingCtx.set_v_dense(self, V)

    def factorize(self, niter, *a, **kw):
        import time
        self.log.append("factorize(%d)" % niter)
        time.sleep(0.05)                                      # (the digest thread finishes first: its abort is on record)
        return _CountingCtx.factorize(self, niter, *a, **kw)


def test_late_data_check_restarts_the_loop_when_the_bytes_changed():
    """Round 5: the digest of `data` runs on a second thread beside the device loop.  Unchanged bytes: snapshots, one loop,
    nothing else.  Bytes edited in place since the upload: the loop is asked to stop, W and H are put back from the device-side
    copies, the new bytes go up, the loop runs again -- and the result is what a check-first object computes."""
    import oracle
    rs = np.random.RandomState(8)
    V = rs.rand(64, 16).astype(np.float32)
    W0, H0 = rs.rand(64, 3), rs.rand(3, 16)

    def model(late):
        mdl = pymf_amd.NMF(V.copy(), num_bases=3)
        mdl._ctx = _LateCtx(64, 16, 3)
        mdl._LATE_DATA_CHECK, mdl._LATE_DATA_CHECK_MIN_BYTES = late, 0
        mdl.W, mdl.H = W0.copy(), H0.copy()
        return mdl
    a, b = model(True), model(False)
    for mdl in (a, b):
        mdl.factorize(niter=2, compute_err=False)             # the first call uploads: checked in front, in both
    assert "snapshot_w" not in a._ctx.log
    a._ctx.log[:] = []
    a.factorize(niter=2, compute_err=False); b.factorize(niter=2, compute_err=False)
    # (the trailing abort(0): however the call ends, the context's abort request is withdrawn -- round-5 advisor: a flag left
    #  set by an interrupted call made the NEXT loop return at once)
    assert a._ctx.log == ["snapshot_w", "snapshot_h", "factorize(2)", "abort(0)"], a._ctx.log
    a._ctx.log[:] = []
    for mdl in (a, b):
        mdl.data[5, 3] += 0.5                                 # in place: same object, new bytes
    a.factorize(niter=3, compute_err=False); b.factorize(niter=3, compute_err=False)
    assert a._ctx.log == ["snapshot_w", "snapshot_h", "factorize(3)", "abort(1)", "abort(0)", "restore_w", "restore_h", "set_v", "factorize(3)", "abort(0)"] or \
        a._ctx.log == ["snapshot_w", "snapshot_h", "abort(1)", "factorize(3)", "abort(0)", "restore_w", "restore_h", "set_v", "factorize(3)", "abort(0)"], a._ctx.log
    assert not a._ctx.aborted and a._ctx.up["V"] == 2 == b._ctx.up["V"]
    np.testing.assert_array_equal(a._ctx.W, b._ctx.W)
    np.testing.assert_array_equal(a._ctx.H, b._ctx.H)
    np.testing.assert_array_equal(a.W, b.W)


def test_an_interrupted_call_does_not_leave_the_abort_request_behind():
    """Round-5 advisor: pmf_abort is sticky.  The digest thread sets it when `data` changed under the running loop; if the main
    thread then leaves factorize() without reaching the restart (a KeyboardInterrupt behind the loop), the flag stayed set and
    the NEXT pmf_factorize returned at once with iters_done = 0 and no error.  Now the call ends -- however it ends -- with the
    digest thread joined and the request withdrawn."""
    rs = np.random.RandomState(9)
    V = rs.rand(64, 16).astype(np.float32)

    class Interrupted(_LateCtx):
        boom = False

        def factorize(self, niter, *a, **kw):
            out = _LateCtx.factorize(self, niter, *a, **kw)
            if self.boom:
                raise KeyboardInterrupt()
            return out
    mdl = pymf_amd.NMF(V, num_bases=3)
    mdl._ctx = Interrupted(64, 16, 3)
    mdl._LATE_DATA_CHECK, mdl._LATE_DATA_CHECK_MIN_BYTES = True, 0
    mdl.W, mdl.H = rs.rand(64, 3), rs.rand(3, 16)
    mdl.factorize(niter=2, compute_err=False)                 # the first call uploads
    V[3, 3] += 1.0                                            # edited in place: the digest thread will ask the loop to stop ...
    mdl._ctx.boom = True
    mdl._ctx.log[:] = []
    with pytest.raises(KeyboardInterrupt):                    # ... and the main thread never gets to the restart
        mdl.factorize(niter=3, compute_err=False)
    assert "abort(1)" in mdl._ctx.log and mdl._ctx.log[-1] == "abort(0)", mdl._ctx.log
    assert mdl._ctx.aborted is False
