#!/bin/bash
# Per-kernel rocprofv3 times of a script (GPU box): [ENV=...] scripts/prof_script.sh <outname> <script.py>   -> gpurun_out/<outname>/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/$2 > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
tail -1 $OUT.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms over %d kernels" % (tot / 1e6, len(rows)))
for r in rows[:28]:
    print("   %-70s calls %5s avg %9.1f us  total %8.2f ms %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
PY
