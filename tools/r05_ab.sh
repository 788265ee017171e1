#!/bin/bash
# old (HEAD) vs new pmf_fused.h through tools/fused_ab.hip: W and slab checksums must agree (same summation order), time per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for n in 44:131072 44:1048576 42:262144 16:1048576 25:262144; do
  k=${n%%:*}; m=${n##*:}
  for rep in 1 2; do for v in old new; do echo "-- $v <$k> m=$m"; build_ab/fab_${v}_$k $m | tail -2; done; done
done
