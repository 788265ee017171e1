#!/usr/bin/env python3
"""Summarise the rocprofv3 counter_collection CSVs of tools/pmc_configs.sh: mean per dispatch of every
counter for every kernel and config -> <dir>/<tag>_pmc_summary.csv, plus the HBM traffic per launch of
each config's dominant kernel (2 * FETCH_SIZE + WRITE_SIZE in bytes: MI355X_MICROARCH.md 'HBM' --
on gfx950 FETCH_SIZE reports half of a wide coalesced streaming read) -> <dir>/<tag>_traffic.json."""
import collections, csv, glob, json, os, sys

O, tag = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(lambda: collections.defaultdict(list))      # (cfg, kernel) -> counter -> values
for d in sorted(glob.glob(os.path.join(O, "cfg*_*"))):
    if not os.path.isdir(d):
        continue
    cfg = os.path.basename(d).split("_")[0]
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[(cfg, r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(O, "%s_pmc_summary.csv" % tag), "w") as fh:
    fh.write("config,kernel,counter,dispatches,mean_per_dispatch\n")
    for (cfg, kern) in sorted(rows):
        if kern.startswith("__amd") or "fill" in kern:
            continue
        for c in sorted(rows[(cfg, kern)]):
            v = rows[(cfg, kern)][c]
            fh.write('%s,"%s",%s,%d,%.3f\n' % (cfg, kern, c, len(v), sum(v) / len(v)))
traffic = {}
for (cfg, kern), d in rows.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f, w = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"]), sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        traffic["%s|%s" % (cfg, kern.split("(")[0])] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                                                        "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024}
json.dump(traffic, open(os.path.join(O, "%s_traffic.json" % tag), "w"), indent=1, sort_keys=True)
big = sorted(traffic.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]
for k, v in big:
    print("%-90s %10.1f MB/launch" % (k[:90], v["hbm_bytes_per_launch"] / 1e6))
