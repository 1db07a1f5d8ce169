#!/bin/bash
# Per-kernel rocprofv3 times of the headline step for library variants (GPU box): [RAYS=4096] scripts/prof_bench_variants.sh <suffix>...  ("-" = default)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "-" ]; then unset UCNERF_LIB; else export UCNERF_LIB=$R/build/variants/libucnerf_hip_$v.so; fi
  rm -rf $R/gpurun_out/bv_$v
  [ -n "$RAYS" ] && echo "-- rays per step: $RAYS"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/bv_$v -- python3 $R/bench.py --rays ${RAYS:-4096} --steps 100 --warmup 20 --no-reuse --cpu-rays 0 > $R/gpurun_out/bv_$v.log 2>&1 || exit 1
  echo "== $v"
  python3 - "$R/gpurun_out/bv_$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("   %-60s calls %4s avg %9.1f us  min %8.1f max %8.1f  %5.1f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
PY
done
