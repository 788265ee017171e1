#!/usr/bin/env python3
"""Hold every literal tolerance of the GPU parity tests next to what was achieved: from the parity ledger
(gpurun_out/parity_r02.json, written by tests/conftest.py) take, per source line and quantity, the worst
achieved error over all parametrisations and rewrite a LITERAL tolerance that is looser than 3x that value
to 3x (rounded up to one significant digit, floor 1e-9).  The named tolerances TOL_X = 2e-5 / TOL_F = 1e-5 (the
stated bar of SURVEY 8(d)) are replaced site by site the same way -- never loosened; computed tolerances and
comparisons with an atol are left alone and listed for a manual look."""
import collections, json, math, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ledger = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_r02.json")))["entries"]
worst = collections.defaultdict(float)
for e in ledger:
    worst[(e["at"], e["what"])] = max(worst[(e["at"], e["what"])], e["achieved"])


def round_up(x):
    x = max(x, 1e-9)
    p = 10.0 ** math.floor(math.log10(x))
    return math.ceil(x / p - 1e-9) * p


LIT = r"(\d+(?:\.\d+)?e-?\d+|\d+\.\d+|TOL_X|TOL_F)"
NAMED = {"TOL_X": 2e-5, "TOL_F": 1e-5}
files = collections.defaultdict(dict)
for (at, what), ach in worst.items():
    f, line = at.split(":")
    files[f][(int(line), what)] = ach
changed, skipped = 0, []
for f, sites in files.items():
    path = os.path.join(ROOT, "tests", f)
    lines = open(path).read().split("\n")
    for (ln, what), ach in sorted(sites.items()):
        src = lines[ln - 1]
        target = round_up(3.0 * ach)
        w = re.escape('what="%s"' % what)
        done = False
        for pat in (w + r"\)\s*<=?\s*" + LIT, r"rtol=" + LIT + r",\s*" + w + r"\)"):
            m = re.search(pat, src)
            if m and "atol" not in src:
                old = NAMED.get(m.group(1)) or float(m.group(1))
                if old > target:
                    new = ("%.0e" % target).replace("e-0", "e-").replace("e+00", "")
                    lines[ln - 1] = src[:m.start(1)] + new + src[m.end(1):]
                    changed += 1
                done = True
                break
        if not done:
            skipped.append((f, ln, what, ach))
    open(path, "w").write("\n".join(lines))
print("tightened %d literal tolerances" % changed)
for s in skipped:
    print("left alone (named / computed / with atol): %s:%d %s achieved %.2g" % s)
