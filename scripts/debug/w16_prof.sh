cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/w16prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/w16prof -- python3 $R/scripts/debug/w16_check.py > $R/gpurun_out/w16prof.log 2>&1
python3 - "$R/gpurun_out/w16prof" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    d[(k, r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (k, g), v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) > 50: print("%-62s grid %8s calls %5d avg %8.2f min %8.2f" % (k, g, len(v), sum(v) / len(v), min(v)))
PY
