#!/usr/bin/env python3
"""usage: trace_durations.py <rocprofv3 kernel_trace.csv> <substring> -- per-launch durations (us) of matching kernels, in launch order"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    print("%-28s %10.1f us  grid %s" % (r["Kernel_Name"][:28], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]))
