#!/bin/bash
# usage: tools/pmc_kernel.sh <out dir under gpurun_out> <kernel substring> -- <python script + args>
# rocprofv3 PMC pass over a script; prints, per dispatch of the matching kernel (launch order), the counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; K=$2; shift 3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PMC=${PMC:-SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE}   # PMC="..." overrides the counter list
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $O/pmc -- python3 "$@" > /dev/null 2>&1
python3 - $O "$K" <<'PY'
import csv, glob, sys, collections
O, K = sys.argv[1], sys.argv[2]
d = collections.OrderedDict()
for f in glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if K in r["Kernel_Name"]:
            d.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for k in sorted(d):
    print(k, " ".join("%s=%.3g" % kv for kv in sorted(d[k].items())))
PY
