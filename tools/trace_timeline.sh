#!/bin/bash
# usage: tools/trace_timeline.sh <out name under gpurun_out> <last N dispatches to keep> -- <python script + args>
# rocprofv3 kernel trace of a script; keeps a compact timeline (start, end in us relative to the first kept dispatch, queue, kernel) of the
# LAST N dispatches -- the steady state of a loop -- as <out>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; N=$2; shift 3
mkdir -p $O.d
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O.d/tr -- python3 "$@" > /dev/null 2>&1
python3 - $O $N <<'PY'
import csv, glob, sys
O, N = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(O + ".d/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
import os
anchor = os.environ.get("ANCHOR")          # ANCHOR="kernel substring:occurrence": N dispatches from that one on (default: the last N)
if anchor:
    name, occ = anchor.rsplit(":", 1)
    hits = [i for i, r in enumerate(rows) if name in r[3]]
    i0 = hits[min(int(occ), len(hits) - 1)] if hits else 0
    rows = rows[i0:i0 + N]
else:
    rows = rows[-N:]
t0 = rows[0][0]
with open(O + ".txt", "w") as fh:
    for s, e, q, k in rows:
        fh.write("%10.1f %10.1f %8.1f  q%s  %s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, k))
print(open(O + ".txt").read()[-6000:])
PY
rm -rf $O.d
