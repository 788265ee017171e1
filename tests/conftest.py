import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    if "V" not in d and "V_seed" in d:
        m, n = (int(x) for x in d["V_shape"])
        u = np.random.RandomState(int(d["V_seed"])).random_sample((m, n))
        # seed 5 is the binary BNMF matrix (gen_golden.py), every other seeded V is U[0,1)
        if "V_shift" in d:                 # gen_golden.py seeded(): (u - shift) rounded to float32
            u = u - float(d["V_shift"])
        d["V"] = (u < 0.2).astype(np.float32) if name.startswith("bnmf") else u.astype(np.float32)
    return d


@pytest.fixture
def golden():
    return load_golden


# ---- parity ledger -------------------------------------------------------------------------------
# Every comparison of a device result with the oracle / a reference golden goes through rel_fro()
# or close(): both record (test, source line, quantity, achieved error, tolerance) and the session
# writes them to gpurun_out/parity_r03.json (copied to profiles/ and committed), so the achieved
# errors -- not only "it passed" -- are on record and the tolerances can be held next to them.
_LEDGER = []
_CURRENT = {"test": None}
LEDGER_PATH = os.environ.get("PMF_PARITY_LEDGER", os.path.join(ROOT, "gpurun_out", "parity_r03.json"))


@pytest.fixture(autouse=True)
def _ledger_current_test(request):
    _CURRENT["test"] = request.node.nodeid
    yield
    _CURRENT["test"] = None


def _record(kind, what, value, tol, depth=2):
    f = sys._getframe(depth)
    _LEDGER.append({"test": _CURRENT["test"], "at": "%s:%d" % (os.path.basename(f.f_code.co_filename), f.f_lineno),
                    "kind": kind, "what": what, "achieved": float(value), "tol": float(tol),
                    "ok": bool(value <= tol)})


class Measured(float):
    """A float that remembers what it measures; comparing it with a tolerance records the pair."""
    what = ""

    def __lt__(self, tol):
        _record("rel_fro", self.what, float(self), float(tol))
        return float(self) < float(tol)

    def __le__(self, tol):
        _record("rel_fro", self.what, float(self), float(tol))
        return float(self) <= float(tol)


def rel_fro(a, b, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = Measured(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    m.what = what
    return m


def close(actual, desired, rtol=1e-7, atol=0.0, what="", err_msg=""):
    """np.testing.assert_allclose that also records the achieved worst-case error in units of the
    tolerance band: max |a - d| / (atol + rtol |d|) * rtol  (the relative error when atol = 0)."""
    a = np.asarray(actual, dtype=np.float64)
    d = np.asarray(desired, dtype=np.float64)
    if a.shape == d.shape and a.size:
        band = atol + rtol * np.abs(d)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(band > 0, np.abs(a - d) / band, np.where(a == d, 0.0, np.inf))
        _record("allclose", what, float(np.nanmax(r)) * rtol, rtol)
    np.testing.assert_allclose(actual, desired, rtol=rtol, atol=atol, err_msg=err_msg)


def pytest_sessionfinish(session, exitstatus):
    if not _LEDGER:
        return
    import json
    try:
        os.makedirs(os.path.dirname(LEDGER_PATH), exist_ok=True)
        with open(LEDGER_PATH, "w") as f:
            json.dump({"entries": _LEDGER}, f, indent=0)
    except OSError:
        pass
