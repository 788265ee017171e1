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
