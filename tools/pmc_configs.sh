#!/bin/bash
# usage: tools/pmc_configs.sh <out tag> [cfg ...]  -- rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in passes of their
# own, MI355X_MICROARCH.md 'rocprofv3 PMC slots') over bench.py for each BASELINE config; summary by tools/pmc_summary.py
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r03}; shift
cfgs=${@:-cfg4 cfg2 cfg3 cfg5}
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
declare -A ARGS=( [cfg4]="--steps 10 --warmup 2 --fill device" [cfg2]="--config cfg2 --steps 20 --warmup 2 --fill device" \
                  [cfg3]="--config cfg3 --steps 3 --warmup 1 --fill device" [cfg5]="--config cfg5 --steps 10 --warmup 2 --fill fast" )
for c in $cfgs; do
  for pass in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/${c}_$name -- python3 $R/bench.py --gpus 1 ${ARGS[$c]} --no-cpu-baseline --preroll-ms 0 > $O/${c}_$name.log 2>&1
  done
done
python3 $R/tools/pmc_summary.py $O $tag
