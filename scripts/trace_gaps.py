"""Per-kernel durations AND the idle gaps between consecutive launches of a rocprofv3 --kernel-trace run, over the last `tail` launches:
  python3 scripts/trace_gaps.py <dir with *_kernel_trace.csv> [tail=3000] [out.json]
Prints, per kernel name, calls / average duration / average gap to the previous kernel's end, and the busy fraction of the window."""
import collections
import csv
import glob
import json
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))))
rows = rows[-tail:]
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for i, (s, e, k) in enumerate(rows):
    dur[k].append(e - s)
    if i:
        gap[k].append(max(0, s - rows[i - 1][1]))
window = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
res = {"trace": f.split("/")[-1], "launches": len(rows), "window_us": window / 1e3, "busy_frac": busy / window, "kernels": {}}
print("%-58s %6s %10s %10s" % ("kernel", "calls", "avg us", "gap-before us"))
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = dur[k], gap.get(k, [0])
    res["kernels"][k] = {"calls": len(d), "avg_us": sum(d) / len(d) / 1e3, "avg_gap_before_us": sum(g) / len(g) / 1e3}
    print("%-58s %6d %10.2f %10.2f" % (k[:58], len(d), sum(d) / len(d) / 1e3, sum(g) / len(g) / 1e3))
print("busy fraction of the window: %.3f (%.1f us of launches in %.1f us)" % (busy / window, busy / 1e3, window / 1e3))
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
