#!/bin/bash
# usage: tools/trace_kernel.sh <out dir under gpurun_out> <kernel substring> -- <python script + args>
# rocprofv3 kernel trace of a script; prints the duration (us) of every dispatch of the matching kernel in launch order
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; K=$2; shift 3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 "$@" > /dev/null 2>&1
python3 - $O "$K" <<'PY'
import csv, glob, sys
O, K = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(O + "/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")))
rows.sort()
print(" ".join("%.0f(g%s)" % (d, g) for _, d, g in rows))
PY
find $O/tr -name "*kernel_trace.csv" -delete
