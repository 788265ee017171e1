#!/bin/bash
# Can two RCCL ranks (two processes) share the ONE GPU of this pool's boxes?  
# Launches the class-level multi-rank worker with the default (RCCL) transport, both ranks on device 0.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-rccl_probe}
mkdir -p $O
cd $R
PORT=29617
for r in 0 1; do
  RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT OMP_NUM_THREADS=1 NCCL_DEBUG=WARN \
    timeout 180 python3 tests/_dist_class_worker.py > $O/rank$r.log 2>&1 &
done
wait
for r in 0 1; do echo "== rank $r"; tail -n 25 $O/rank$r.log; done
