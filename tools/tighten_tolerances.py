#!/usr/bin/env python3
"""Per-case tolerances for the GPU parity tests.

The literal tolerance in a test is the bar of the whole (often parametrised) function.  From parity ledgers --
gpurun_out/parity_r03.json, written by tests/conftest.py on the MI355X boxes; several may be given and the worst
error counts -- this writes tests/golden/tolerances.json: for every single comparison (test id, quantity, n-th
occurrence) TEN times the worst error it has shown (one significant digit, rounded up), never above the stated
literal and never below a floor (1e-9; 1e-12 where the stated tolerance itself is below 1e-8: float64 quantities).
tests/conftest.py applies the smaller of the literal and the table entry.  Ten times, not three: grids derived
from the CU count change summation orders from box to box.

    python tools/tighten_tolerances.py [ledger.json ...]
"""
import json, math, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = sys.argv[1:] or [os.path.join(ROOT, "gpurun_out", "parity_r03.json")]
worst, stated = {}, {}
for p in paths:
    for e in json.load(open(p))["entries"]:
        key = e.get("case")
        if not key:
            continue
        worst[key] = max(worst.get(key, 0.0), e["achieved"])
        stated[key] = e.get("stated_tol", e["tol"])


def round_up(x):
    p = 10.0 ** math.floor(math.log10(x))
    return math.ceil(x / p - 1e-9) * p


cases, tighter = {}, 0
for key, ach in sorted(worst.items()):
    lit = stated[key]
    floor = 1e-12 if lit < 1e-8 else 1e-9
    t = min(lit, round_up(max(10.0 * ach, floor)))
    cases[key] = t
    tighter += t < lit
out = os.path.join(ROOT, "tests", "golden", "tolerances.json")
json.dump({"source": [os.path.basename(p) for p in paths], "rule": "min(literal, roundup(max(10 x worst achieved, floor)))",
           "cases": cases}, open(out, "w"), indent=0, sort_keys=True)
print("%d comparisons, %d of them now tighter than their literal -> %s" % (len(cases), tighter, out))
