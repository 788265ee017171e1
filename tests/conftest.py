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
# writes them to gpurun_out/parity_r06.json (copied to profiles/ and committed), so the achieved
# errors -- not only "it passed" -- are on record and the tolerances can be held next to them.
_LEDGER = []
_CURRENT = {"test": None, "seen": {}}

# Per-case tolerances: the literal in a test is the bar of its whole function (often a parametrised one); the
# table holds, for every single comparison -- (test id, quantity, n-th occurrence) --, ten times the worst error
# that comparison has shown on the MI355X boxes (never above the literal, never below 1e-9 / 1e-12 for float64
# quantities): tools/tighten_tolerances.py writes it from the ledger.  A comparison uses the smaller of the two.
_TOL_TABLE = {}
try:
    if os.environ.get("PMF_NO_TOL_TABLE"):       # (a ledger run after a numerical change: literals only, the table is rebuilt from it)
        raise OSError("per-case table switched off")
    import json as _json
    with open(os.path.join(ROOT, "tests", "golden", "tolerances.json")) as _f:
        _TOL_TABLE = _json.load(_f)["cases"]
except (OSError, ValueError, KeyError):
    _TOL_TABLE = {}


def _case_tol(what, tol):
    key0 = "%s|%s" % (_CURRENT["test"], what)
    n = _CURRENT["seen"].get(key0, 0)
    _CURRENT["seen"][key0] = n + 1
    key = "%s|%d" % (key0, n)
    t = _TOL_TABLE.get(key)
    return (min(float(tol), float(t)) if t is not None else float(tol)), key
LEDGER_PATH = os.environ.get("PMF_PARITY_LEDGER", os.path.join(ROOT, "gpurun_out", "parity_r06.json"))


@pytest.fixture(autouse=True)
def _ledger_current_test(request):
    _CURRENT["test"] = request.node.nodeid
    _CURRENT["seen"] = {}
    yield
    _CURRENT["test"] = None


def _record(kind, what, value, tol, depth=2, stated=None, key=None):
    f = sys._getframe(depth)
    _LEDGER.append({"test": _CURRENT["test"], "at": "%s:%d" % (os.path.basename(f.f_code.co_filename), f.f_lineno),
                    "kind": kind, "what": what, "achieved": float(value), "tol": float(tol),
                    "stated_tol": float(tol if stated is None else stated), "case": key,
                    "ok": bool(value <= tol)})


class Measured(float):
    """A float that remembers what it measures; comparing it with a tolerance records the pair."""
    what = ""

    def __lt__(self, tol):
        eff, key = _case_tol(self.what, tol)
        _record("rel_fro", self.what, float(self), eff, stated=tol, key=key)
        return float(self) < eff

    def __le__(self, tol):
        eff, key = _case_tol(self.what, tol)
        _record("rel_fro", self.what, float(self), eff, stated=tol, key=key)
        return float(self) <= eff


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
    stated = rtol
    if atol == 0.0:
        rtol, key = _case_tol(what, rtol)          # (a band with an absolute part keeps its literal)
    else:
        key = None
    if a.shape == d.shape and a.size:
        band = atol + stated * np.abs(d)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = np.where(band > 0, np.abs(a - d) / band, np.where(a == d, 0.0, np.inf))
        _record("allclose", what, float(np.nanmax(r)) * stated, rtol, stated=stated, key=key)
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
