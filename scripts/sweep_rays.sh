#!/bin/bash
# Per-kernel times of the headline step at four batch sizes (GPU box): bash scripts/sweep_rays.sh
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for R in 1024 2048 4096 8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_$R -o s -- python3 bench.py --rays $R --steps 100 --warmup 20 --no-reuse --cpu-rays 0 > gpurun_out/sweep_$R.log 2>&1
  f=$(find gpurun_out/sweep_$R -name '*kernel_stats.csv' | head -1)
  echo "== $R"; python3 - "$f" <<'P'
import csv,sys
for x in list(csv.DictReader(open(sys.argv[1])))[:6]:
    print(x['Name'][:60].ljust(60), x['Calls'], x['AverageNs'], x['MinNs'], x['MaxNs'])
P
done
