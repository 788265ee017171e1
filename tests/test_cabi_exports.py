"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/pymf_hip.h declares, and the product path has NO CPU fallback."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pymf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmf_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from pymf_amd.csrc import build
    build.build(force=False, verbose=False)       # hipcc cross-compiles without a GPU
    from pymf_amd import _lib
    return _lib


def test_header_declares_the_hot_path_entry_points():
    syms = _declared_symbols()
    for must in ("pmf_ctx_create", "pmf_ctx_destroy", "pmf_set_v_dense_f32", "pmf_set_v_csr_f32",
                 "pmf_set_w_f32", "pmf_get_w_f32", "pmf_set_h_f32", "pmf_get_h_f32", "pmf_update_w",
                 "pmf_update_h", "pmf_frobenius", "pmf_factorize", "pmf_last_error",
                 "pmf_nccl_unique_id", "pmf_device_count"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(raw, name), "libpymf_hip.so does not export %s" % name


def test_python_binding_covers_every_declared_symbol(lib):
    lib.load()
    bound = {name for name, _, _ in lib.SYMBOLS}
    assert bound == set(_declared_symbols())


def test_no_cpu_fallback(lib):
    """Without a GPU the product path must fail loudly, never compute on the host."""
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    import pymf_amd
    mdl = pymf_amd.NMF(np.ones((8, 6), dtype=np.float32), num_bases=2)
    with pytest.raises(lib.PmfError):
        mdl.factorize(niter=1)
    with pytest.raises(lib.PmfError):
        mdl.frobenius_norm()                          # W/H exist now: also needs the device


def test_product_package_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under pymf_amd/ may import it."""
    pkg = os.path.join(ROOT, "pymf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "import torch" not in txt, f        # north_star: no PyTorch in the product
    # ... nor may the diagnostics under tools/ (the checkers that compare with the oracle live in tests/sweeps/); the only
    # users outside tests/ are __graft_entry__.smoke() and bench.py's cpu_baseline / parity legs
    for f in os.listdir(os.path.join(ROOT, "tools")):
        if f.endswith(".py"):
            txt = open(os.path.join(ROOT, "tools", f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), "tools/" + f


def _create_code(lib, *args):
    try:
        lib.Context(*args).close()
    except lib.PmfError as e:
        return e.code
    return lib.PMF_OK


def test_bad_arguments_return_einval(lib):
    """pmf_ctx_create validates its arguments BEFORE it touches a device, so the status is PMF_EINVAL
    with or without a GPU -- and a valid shape without a GPU is PMF_EHIP, not EINVAL."""
    lib.load()
    assert (lib.PMF_OK, lib.PMF_EINVAL, lib.PMF_EHIP, lib.PMF_ENCCL, lib.PMF_ENOMEM, lib.PMF_ESINGULAR) == (0, -1, -2, -3, -4, -5)
    assert _create_code(lib, lib.ALGO_NMF, 0, 4, 2) == lib.PMF_EINVAL          # m < 1
    assert _create_code(lib, lib.ALGO_NMF, 4, 0, 2) == lib.PMF_EINVAL          # n < 1
    assert _create_code(lib, lib.ALGO_NMF, 4, 4, 0) == lib.PMF_EINVAL          # k < 1
    assert _create_code(lib, 7, 4, 4, 2) == lib.PMF_EINVAL                     # unknown algo
    assert _create_code(lib, lib.ALGO_NMF, 4, 4, 2433) == lib.PMF_EINVAL       # num_bases beyond the build's limit
    assert _create_code(lib, lib.ALGO_NMFALS, 4, 4, 1025) == lib.PMF_EINVAL    # (NMFALS / NMFNNLS: 1 024)
    assert _create_code(lib, lib.ALGO_NMF, 4, 4, 2, 0, 3, 2) == lib.PMF_EINVAL  # rank >= nranks
    assert _create_code(lib, lib.ALGO_NMF, 4, 4, 2, 0, 0, 2, None) == lib.PMF_EINVAL   # nranks > 1 without an id
    valid = _create_code(lib, lib.ALGO_NMF, 4, 4, 129)                         # num_bases > 128 is a supported NMF shape
    if lib.device_count() > 0:
        assert valid == lib.PMF_OK
    else:
        assert valid == lib.PMF_EHIP                                           # no device: a HIP error, never EINVAL


@pytest.mark.gpu
def test_bad_arguments_on_the_gpu_box(lib):
    """The same status codes with a device present, plus the call-order errors of a live context."""
    test_bad_arguments_return_einval(lib)
    ctx = lib.Context(lib.ALGO_NMF, 8, 6, 2)
    for call in (ctx.update_w, ctx.update_h, ctx.frobenius, lambda: ctx.factorize(1)):
        with pytest.raises(lib.PmfError) as ei:
            call()                                                             # V / W / H not set yet
        assert ei.value.code == lib.PMF_EINVAL
    ctx.close()


def test_host_checksum_is_position_sensitive(lib):
    """The change detector of the host classes (pmf_host_checksum, no device needed): any in-place
    edit -- sum-preserving ones included -- must change the digest; equal bytes must not."""
    rs = np.random.RandomState(0)
    W = rs.random_sample((5000, 64))
    base = lib.host_checksum(W)
    assert lib.host_checksum(W.copy()) == base
    X = W.copy(); X[[3, 4000]] = X[[4000, 3]]                  # swap two rows: same sum, same last element
    assert lib.host_checksum(X) != base
    X = W.copy(); X[0, 0] += 1.0; X[1, 0] -= 1.0               # sum-preserving poke
    assert lib.host_checksum(X) != base
    X = W.copy(); X[:] = X[:, ::-1]                            # permute the bases in place
    assert lib.host_checksum(X) != base
    X = W.copy(); X[17, 5] = np.nextafter(X[17, 5], 2.0)       # one ulp
    assert lib.host_checksum(X) != base
    assert lib.host_checksum(W.astype(np.float32)) != base
    assert lib.host_checksum(W.reshape(64, 5000))[2:] == base[2:]      # same bytes, other shape: digest equal ...
    assert lib.host_checksum(W.reshape(64, 5000)) != base              # ... the shape is part of the fingerprint
    big = rs.random_sample(5 * (1 << 20) + 13).astype(np.float32)      # several 8 MiB chunks + a ragged tail: threaded path
    d0 = lib.host_checksum(big)
    assert lib.host_checksum(big.copy()) == d0
    big[3 * (1 << 20) + 7], big[3 * (1 << 20) + 8] = big[3 * (1 << 20) + 8], big[3 * (1 << 20) + 7]
    assert lib.host_checksum(big) != d0
    nc = W[::2, ::3]                                                   # non-contiguous view: digest of its contents
    assert lib.host_checksum(nc) == lib.host_checksum(np.ascontiguousarray(nc))
    assert lib.host_checksum(np.zeros((0, 4)))[0] == (0, 4)


@pytest.mark.gpu
def test_every_documented_option_is_known_to_the_library(lib):
    """include/pymf_hip.h lists the tuning knobs of pmf_set_option by name: each must be accepted with its default
    value, an undocumented name must be PMF_EINVAL."""
    import re
    from pymf_amd import _lib
    text = open(os.path.join(ROOT, "include", "pymf_hip.h")).read()
    block = text[text.index("Tuning knobs"):text.index("int pmf_set_option")]
    names = re.findall(r'^ \*   "([a-z0-9_]+)"', block, flags=re.M)
    assert set(names) >= {"snmf_gram", "nnqp_quad", "force_tiled", "rowgemm_stream", "colgemm_stream", "nndsvd_topk", "nnqp_frame16"}, names
    defaults = {"snmf_gram": -1, "nnqp_quad": 1, "force_tiled": 0, "rowgemm_stream": 1, "colgemm_stream": 1, "nndsvd_topk": -1, "nnqp_frame16": 1, "profile_every": 1, "fold_exchange": 1}
    ctx = _lib.Context(_lib.ALGO_NMF, 256, 64, 8)
    for name in names:
        ctx.set_option(name, defaults.get(name, 0))
    with pytest.raises(_lib.PmfError):
        ctx.set_option("no_such_option", 1)
    ctx.close()


def test_a_library_built_from_other_sources_is_refused(monkeypatch):
    """the driver's box loads the .so that travelled with the snapshot; a binary that does not belong to the
    sources at hand must not pass silently.  build.py stamps the library with a SHA-256 of csrc/*.h, csrc/*.hip,
    include/pymf_hip.h and the flags; _lib.load() compares."""
    from pymf_amd import _lib
    from pymf_amd.csrc import build as b
    assert b.up_to_date() and b.built_hash() == b.source_hash()          # the tree under test is consistent
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.delenv("PMF_LIB", raising=False)
    monkeypatch.setattr(b, "source_hash", lambda: "0" * 64)
    with pytest.raises(_lib.PmfError, match="not built from the sources at hand"):
        _lib.load()
    monkeypatch.undo()
    assert _lib.load() is not None


# Kernels allowed to keep a private (scratch) segment, with its size in bytes per lane.  k_snmf_csr_mfma<8, 8>: round 1's
# one-pass CSR kernel at 128 bases -- 512 registers, three spilled dwords; not selected by default (the Gram-space loop and
# k_csr_w_blocks carry cfg5), kept for the pass-per-iteration A/B in bench.py.
SCRATCH_ALLOWED = {"k_snmf_csr_mfma<8, 8>": 12}


def test_no_product_kernel_uses_scratch(lib):
    """k_nmf_h_gram<8,*> picked up 88-120 B of scratch per lane when the folded exchange's arguments
    arrived and nobody noticed.  Every gfx950 kernel of the built library is read back (AMDGPU metadata notes of the code
    objects: .private_segment_fixed_size) and a non-zero entry outside the allow-list above fails the suite."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_regs
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf of the ROCm image is not here")
    rows = kernel_regs.kernels(lib.LIB_PATH)
    assert len(rows) > 200, "the metadata of the code objects was not found (%d kernels)" % len(rows)
    bad = [(r["dem"], r["scratch"]) for r in rows
           if int(r["scratch"]) != 0 and SCRATCH_ALLOWED.get(r["dem"]) != int(r["scratch"])]
    assert not bad, "kernels with a private segment (spills): %s" % bad
    # the instantiations the verdict named, by name: both forms of the 128-base H step, folded exchange or not
    names = {r["dem"] for r in rows}
    for want in ("k_nmf_h_gram<8, false, false>", "k_nmf_h_gram<8, true, false>", "k_nmf_h_gram<8, false, true>", "k_nmf_h_gram<8, true, true>"):
        assert want in names, want
