#!/bin/bash
# same-box kernel averages of the training-style step: ab/libucnerf_base.so against the tree's library
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in base new base new; do
  rm -rf $R/gpurun_out/abp_$v
  if [ $v = base ]; then export UCNERF_LIB=$R/ab/libucnerf_base.so; else unset UCNERF_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abp_$v -- python3 $R/scripts/time_train_step.py > $R/gpurun_out/abp_$v.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/abp_$v/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
print("$v", " | ".join("%s %.1f"%(r['Name'].split('(')[0].split('::')[-1][:22], float(r['AverageNs'])/1e3) for r in rows[:5]), "| total/13 %.1f us"%(sum(float(r['TotalDurationNs']) for r in rows)/13e3))
PY
done
