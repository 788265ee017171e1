#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...>   -- one rocprofv3 PMC pass over quick_bench (cfg4)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tools/quick_bench.py 1048576 256 64 10 > $R/gpurun_out/pmc_$tag.log 2>&1
f=$(find $R/gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: (len(v), sum(v) / len(v)) for c, v in d.items()})
PY
