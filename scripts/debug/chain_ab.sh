#!/bin/bash
# gradient parity tests + the training-style step under a kernel trace (GPU box)
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_hip_round3.py tests/test_hip_round4.py -m gpu -x -q > gpurun_out/chain_ab_tests.log 2>&1 || { tail -30 gpurun_out/chain_ab_tests.log; exit 1; }
tail -3 gpurun_out/chain_ab_tests.log
python scripts/time_train_step.py
ZC=1 python scripts/time_train_step.py
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/chain_ab_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/chain_ab_prof -- python3 $R/scripts/time_train_step.py > $R/gpurun_out/chain_ab_prof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/chain_ab_prof/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
