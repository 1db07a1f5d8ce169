#!/bin/bash
# Kernel trace of the reference's evaluation loop through the drop-in (one 256 x 320 image = 80 chunks): launches and GPU time per chunk
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/eval_trace
IMAGES=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/eval_trace -- python3 $R/scripts/debug/prof_eval_image.py > $R/gpurun_out/eval_trace.log 2>&1
python3 - $R/gpurun_out/eval_trace <<'PY'
import csv, glob, sys, collections
k = sorted(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
# the last image of the run: the last 80 launches of the gather-fused kernel delimit it
fused = [i for i, r in enumerate(k) if "mlp_fwd_bf16_kernel" in r["Kernel_Name"]]
last = fused[-80:]
lo = max(0, last[0] - 3)
rows = k[lo:last[-1] + 1]
names = collections.Counter(r["Kernel_Name"].split("(")[0].replace("ucnerf::", "").replace("void ", "")[:60] for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("last image: %d launches over %.1f ms, GPU busy %.2f ms (%.1f us per chunk), busy fraction %.2f" % (len(rows), span / 1e3, busy / 1e3, busy / 80, busy / span))
for n, c in names.most_common(8):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].split("(")[0].replace("ucnerf::", "").replace("void ", "")[:60] == n]
    print("  %-62s x%-4d avg %7.1f us" % (n, c, sum(d) / len(d) / 1e3))
PY
