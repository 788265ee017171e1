#!/bin/bash
# usage: tools/run_bench_configs.sh <out tag>   -- every BASELINE config through bench.py, then the same
# commands under rocprofv3 --kernel-trace --stats (summaries for profiles/)
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r04}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
declare -A ARGS=( [cfg4]="--steps 20 --warmup 5" [cfg2]="--config cfg2 --steps 200 --warmup 5" \
                  [cfg3]="--config cfg3 --steps 50 --warmup 3" [cfg5]="--config cfg5 --steps 20 --warmup 5" )
for c in ${CONFIGS:-cfg4 cfg2 cfg3 cfg5}; do
  python3 $R/bench.py --gpus 1 ${ARGS[$c]} > $O/$c.json 2> $O/$c.err
  tail -c 400 $O/$c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python3 $R/bench.py --gpus 1 ${ARGS[$c]} --no-cpu-baseline > $O/${c}_under_rocprof.json 2> $O/${c}_rocprof.err
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${c}_kernel_stats.csv
  find $O/prof_$c -name "*kernel_trace.csv" -delete
done
python3 - $O <<'PY'
import json, sys, glob, os
O = sys.argv[1]
for c in (os.environ.get('CONFIGS') or 'cfg4 cfg2 cfg3 cfg5').split():
    for suffix in ("", "_under_rocprof"):
        p = os.path.join(O, c + suffix + ".json")
        try:
            j = [json.loads(l) for l in open(p) if l.startswith("{")][-1]
        except Exception as e:
            print(c + suffix, "no line:", e); continue
        r = j["roofline"]
        print("%-20s %9.1f it/s %8.4f ms/step | %s mean %.4f min %.4f med %.4f max %.4f ms | %s %.1f %s frac %.3f share %.2f | cpu %s" % (
            c + suffix, j["value"], j["ms_per_step"], r["kernel"], r["mean_kernel_ms"], r["min_kernel_ms"] or 0, r["median_kernel_ms"] or 0,
            r["max_kernel_ms"] or 0, r["bound"], r["achieved"], r["unit"], r["frac"], r["kernel_share_of_timed_region"],
            j.get("cpu_baseline", {}).get("value")))
PY
for c in ${CONFIGS:-cfg4 cfg2 cfg3 cfg5}; do echo "== $c"; head -8 $O/${c}_kernel_stats.csv 2>/dev/null | cut -c1-160; done
