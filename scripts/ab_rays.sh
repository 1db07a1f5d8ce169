#!/bin/bash
# Step time of the two routes (gather fused into the MLP kernel / two-kernel pass) over ray counts (GPU box): scripts/ab_rays.sh 512 1024 ...
R=$(cd "$(dirname "$0")/.." && pwd)
for n in "$@"; do
  for prec in bf16x3 bf16x3_fused; do
    timeout -k 10 120 python $R/bench.py --rays $n --precision $prec --cpu-rays 0 --no-reuse --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('rays $n', '$prec', 'ms/step %.4f' % d['ms_per_step'], 'M rays/s %.3f' % (d['value']/1e6))" || exit 1
  done
done
