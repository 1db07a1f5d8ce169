#!/bin/bash
# Kernel trace of the strong-scaling shard's step (GPU box): [RAYS=512] scripts/prof_strong.sh <tag>   -> gpurun_out/strong_<tag>_{repack,hoisted}.*
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
T=${1:-base}
for rp in 1 0; do
  name=strong_${T}_$([ $rp = 1 ] && echo repack || echo hoisted)
  rm -rf $R/gpurun_out/$name
  REPACK=$rp rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -- python3 $R/scripts/time_strong.py > $R/gpurun_out/$name.log 2>&1 || exit 1
  tail -1 $R/gpurun_out/$name.log
  python3 $R/scripts/trace_gaps.py $R/gpurun_out/$name 3000 $R/gpurun_out/$name.gaps.json
  cp $(ls $R/gpurun_out/$name/*/*kernel_stats.csv | tail -1) $R/gpurun_out/$name.kernel_stats.csv
done
# the untraced timing of the same two steps
REPACK=1 python3 $R/scripts/time_strong.py && REPACK=0 python3 $R/scripts/time_strong.py
