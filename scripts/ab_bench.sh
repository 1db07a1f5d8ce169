#!/bin/bash
# A/B timing of library variants on the GPU box: scripts/ab_bench.sh <precision> <suffix>...   ("-" = the default library)
R=$(cd "$(dirname "$0")/.." && pwd)
PREC=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset UCNERF_LIB; else export UCNERF_LIB=$R/build/variants/libucnerf_hip_$v.so; fi
  timeout -k 10 120 python $R/bench.py --precision $PREC --cpu-rays 0 --no-reuse 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$v', 'ms/step %.3f' % d['ms_per_step'], 'mlp launch ms %.4f' % r['avg_launch_ms'], 'TF %.1f' % r['achieved'], 'share %.3f' % d['mlp_share_of_step'])" || exit 1
done
