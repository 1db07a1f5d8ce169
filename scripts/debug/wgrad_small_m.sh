#!/bin/bash
# The weight-gradient launch at small m: per-kernel times of the training-style step at 250 x 90 and 2000 x 90 for the
# default library and the no-flush timing variant (build it first: SRC=mlp_wgrad scripts/build_variant.sh noflush -DUCNERF_WGRAD_EXP_NOFLUSH=1)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for rays in 250 2000; do
    if [ "$v" = "-" ]; then unset UCNERF_LIB; else export UCNERF_LIB=$R/build/variants/libucnerf_hip_$v.so; fi
    rm -rf $R/gpurun_out/wg_${v}_$rays
    RAYS=$rays SAMPLES=90 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wg_${v}_$rays -- python3 $R/scripts/time_train_step.py > $R/gpurun_out/wg_${v}_$rays.log 2>&1 || exit 1
    echo "== $v $rays: $(grep 'train-style' $R/gpurun_out/wg_${v}_$rays.log)"
    python3 - "$R/gpurun_out/wg_${v}_$rays" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("   %-60s calls %4s avg %9.1f us  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  done
done
