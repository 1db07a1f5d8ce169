#!/bin/bash
# Kernel trace (+ gaps between launches) of the rendering() training step: [RAYS="2000 250"] scripts/prof_dropin_train.sh <tag>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
T=${1:-base}
for n in ${RAYS:-2000 250}; do
  name=dropin_train_${T}_$n
  rm -rf $R/gpurun_out/$name
  RAYS=$n MODE=train STEPS=60 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -- python3 $R/scripts/time_dropin.py > $R/gpurun_out/$name.log 2>&1 || exit 1
  tail -1 $R/gpurun_out/$name.log
  python3 $R/scripts/trace_gaps.py $R/gpurun_out/$name 1500 $R/gpurun_out/$name.gaps.json
  cp $(ls $R/gpurun_out/$name/*/*kernel_stats.csv | tail -1) $R/gpurun_out/$name.kernel_stats.csv
  RAYS=$n MODE=train STEPS=200 python3 $R/scripts/time_dropin.py
done
