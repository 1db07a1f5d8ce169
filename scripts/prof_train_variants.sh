#!/bin/bash
# Per-kernel times of the training-style step for library variants (GPU box): scripts/prof_train_variants.sh <suffix>...  ("-" = default)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "-" ]; then unset UCNERF_LIB; else export UCNERF_LIB=$R/build/variants/libucnerf_hip_$v.so; fi
  rm -rf $R/gpurun_out/tv_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tv_$v -- python3 $R/scripts/time_train_step.py > $R/gpurun_out/tv_$v.log 2>&1 || exit 1
  echo "== $v: $(grep 'train-style' $R/gpurun_out/tv_$v.log)"
  python3 - "$R/gpurun_out/tv_$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print("   %-60s calls %4s avg %9.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
done
