#!/bin/bash
# Step time over ray counts for library variants (GPU box): scripts/ab_rays_lib.sh "<suffix> ..." <rays>...   ("-" = default library; default precision)
R=$(cd "$(dirname "$0")/.." && pwd)
LIBS=$1; shift
for n in "$@"; do
  for v in $LIBS; do
    if [ "$v" = "-" ]; then unset UCNERF_LIB; else export UCNERF_LIB=$R/build/variants/libucnerf_hip_$v.so; fi
    timeout -k 10 120 python $R/bench.py --rays $n --cpu-rays 0 --no-reuse --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('rays $n', 'lib $v', 'ms/step %.4f' % d['ms_per_step'], 'M rays/s %.3f' % (d['value']/1e6))" || exit 1
  done
done
