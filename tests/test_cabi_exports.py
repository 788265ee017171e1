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
                assert "import torch" not in txt or f == "dist.py", f


def test_bad_arguments_return_einval(lib):
    lib.load()
    with pytest.raises(lib.PmfError):
        lib.Context(lib.ALGO_NMF, 0, 4, 2)          # m < 1
    with pytest.raises(lib.PmfError):
        lib.Context(lib.ALGO_NMF, 4, 4, 129)        # num_bases > 128
    with pytest.raises(lib.PmfError):
        lib.Context(7, 4, 4, 2)                     # unknown algo
