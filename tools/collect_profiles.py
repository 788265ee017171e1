#!/usr/bin/env python3
"""Copy the judged rocprofv3 summaries from gpurun_out/ (scratch) into profiles/ (tracked).

usage: tools/collect_profiles.py <round tag, e.g. r01> <stats dir> <bench log> <pmc tags...>
"""
import collections, csv, glob, json, os, shutil, sys

tag, statdir, benchlog = sys.argv[1:4]
pmc_tags = sys.argv[4:]
os.makedirs("profiles", exist_ok=True)
ks = max(glob.glob(os.path.join(statdir, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # newest run
shutil.copy(ks, "profiles/%s_bench_kernel_stats.csv" % tag)
for line in open(benchlog):
    if line.startswith('{"metric"'):
        open("profiles/%s_bench_under_rocprof.json" % tag, "w").write(line)
rows = {}
for t in pmc_tags:
    f = max(glob.glob("gpurun_out/pmc_%s/**/*counter_collection.csv" % t, recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        for c, v in d.items():
            rows.setdefault(k, {})[c] = (len(v), sum(v) / len(v))
with open("profiles/%s_pmc_summary.csv" % tag, "w") as fh:
    fh.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for k in sorted(rows):
        if k.startswith("__amd") or "fill" in k:
            continue
        for c in sorted(rows[k]):
            n, m = rows[k][c]
            fh.write('"%s",%s,%d,%.3f\n' % (k, c, n, m))
fk = [k for k in rows if "k_nmf_fused" in k]
if fk and "FETCH_SIZE" in rows[fk[0]] and "WRITE_SIZE" in rows[fk[0]]:
    fetch_kb, write_kb = rows[fk[0]]["FETCH_SIZE"][1], rows[fk[0]]["WRITE_SIZE"][1]
    # MI355X_MICROARCH.md 'HBM': on gfx950 FETCH_SIZE reports exactly 1/2 of a wide coalesced
    # streaming read -> double it; WRITE_SIZE is exact for 16-B/lane and dword stores.
    traffic = 2 * fetch_kb * 1024 + write_kb * 1024
    json.dump({"k_nmf_fused<4,4>@1048576x256x64/1": traffic,
               "_provenance": "round %s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                              "(tools/pmc_pass.sh): FETCH_SIZE=%.0f KB (x2 gfx950 correction), WRITE_SIZE=%.0f KB "
                              "per dispatch of k_nmf_fused<4,4> at cfg4 on 1 GPU" % (tag, fetch_kb, write_kb)},
              open("profiles/traffic.json", "w"), indent=1)
    print("traffic %.1f MB/launch" % (traffic / 1e6))
