#!/bin/bash
# Step time of the fused route for several values of fused_min_rounds over ray counts (GPU box): scripts/ab_rounds.sh "0 1 2 4" 512 1024 ...
R=$(cd "$(dirname "$0")/.." && pwd)
ROUNDS=$1; shift
for n in "$@"; do
  for r in $ROUNDS; do
    timeout -k 10 120 python $R/bench.py --rays $n --fused-min-rounds $r --cpu-rays 0 --no-reuse --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('rays $n', 'min_rounds $r', 'ms/step %.4f' % d['ms_per_step'], 'M rays/s %.3f' % (d['value']/1e6))" || exit 1
  done
done
