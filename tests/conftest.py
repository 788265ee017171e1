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
        d["V"] = (u < 0.2).astype(np.float32) if name.startswith("bnmf") else u.astype(np.float32)
    return d


@pytest.fixture
def golden():
    return load_golden


def rel_fro(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
