#!/bin/bash
# Per-kernel time of the training-style step for timing variants of the gradient chain (scripts/build_variant.sh, SRC=mlp_bwd_chain):
#   scripts/prof_chain_variants.sh "" chain_exp1 chain_exp2 ...     ("" = the production library)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  L=$R/uc_nerf_amd/libucnerf_hip.so
  [ -n "$v" ] && L=$R/build/variants/libucnerf_hip_$v.so
  rm -rf $R/gpurun_out/pcv
  UCNERF_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pcv -- python3 $R/scripts/time_train_step.py > $R/gpurun_out/pcv.log 2>&1 || { echo "$v: run failed"; tail -3 $R/gpurun_out/pcv.log; continue; }
  python3 - "$v" $R/gpurun_out/pcv/*/*kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if any(k in r["Name"] for k in ("chain", "wgrad", "mlp_fwd_kernel", "gather_bwd", "add_transposed")):
        print("%-14s %-28s %8.1f us" % (sys.argv[1] or "production", r["Name"].split("(")[0][-28:], float(r["AverageNs"]) / 1e3))
PY
done
