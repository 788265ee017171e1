#!/bin/bash
# usage: tools/prof_shape.sh <out dir under gpurun_out> m n k niter algo   -- quick_bench + rocprofv3 kernel stats of one shape
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/quick_bench.py "$@" 2>&1 | grep -v "^path"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/quick_bench.py "$@" > /dev/null 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %5s avg %9.1f us  %5s%%" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'][:5]))
PY
find $O/prof -name "*kernel_trace.csv" -delete
