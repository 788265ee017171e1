#!/bin/bash
# usage: tools/ab_bench.sh <out dir under gpurun_out> [lib ...]   -- A/B builds (build_ab/lib_*.so, selected by PMF_LIB)
# through tools/quick_bench.py on the small-per-rank shapes, then rocprofv3 kernel stats of the cfg2 loop
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
libs=${@:-$(ls $R/build_ab/lib_*.so)}
for lib in $libs; do
  n=$(basename $lib .so)
  echo "=== $n"
  for shape in "65536 512 32 300" "131072 256 64 200" "1048576 256 64 60"; do
    PMF_LIB=$lib timeout 300 python3 $R/tools/quick_bench.py $shape 2>&1 | grep -v "^path"
  done
  PMF_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python3 $R/tools/quick_bench.py 65536 512 32 300 > /dev/null 2>&1
  f=$(find $O/prof_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -6 "$f" | cut -d, -f1-8 | cut -c1-150
  find $O/prof_$n -name "*kernel_trace.csv" -delete
done
