#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/x6prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/x6prof -- python3 $R/scripts/debug/x6_check.py > $R/gpurun_out/x6prof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/x6prof/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'mlp_fwd' in r['Kernel_Name']]
for r in rows[-26:]:
    print(r['Kernel_Name'][:70], r['Grid_Size_X'] if 'Grid_Size_X' in r else '', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
PY
