#!/bin/bash
# kernel trace of the two-stream split -> gpurun_out/two_stream/{log, trace}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/two_stream
for rays in 2000 250; do
  RAYS=$rays python3 $R/scripts/debug/two_stream_bwd.py 2>&1 | grep -v amdgpu.ids | tee -a $R/gpurun_out/two_stream/times.log
done
rm -rf $R/gpurun_out/two_stream/trace
RAYS=2000 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/two_stream/trace -- python3 $R/scripts/debug/two_stream_bwd.py > $R/gpurun_out/two_stream/trace.log 2>&1
python3 - $R/gpurun_out/two_stream/trace <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if any(k in r["Kernel_Name"] for k in ("mlp_bwd_chain", "mlp_wgrad", "mlp_fwd_kernel", "feat_gather_bwd"))]
tail = rows[-16:]                       # the last two-stream step
t0 = int(tail[0]["Start_Timestamp"])
print("last two-stream step (us from its first kernel): name, stream/queue, start, end")
for r in tail:
    print("  %-28s q%-3s %8.1f %8.1f" % (r["Kernel_Name"].split("(")[0].replace("ucnerf::", "")[:28], r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
PY
