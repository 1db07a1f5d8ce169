"""Per-kernel averages of the counters written by scripts/pmc_mlp.sh (any number of distinct kernels): HBM bytes per launch
(FETCH_SIZE x 2: gfx950 tallies 64 B per 128-B request; WRITE_SIZE in KB) and duration."""
import collections
import csv
import glob
import json
import sys

base = sys.argv[1]
out = collections.defaultdict(dict)
for name in ("fetch", "write", "grbm"):
    fs = sorted(glob.glob(base + "/" + name + "/*/*counter_collection.csv"))
    if not fs:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, kname = {}, {}
    for r in csv.DictReader(open(fs[-1])):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        kname[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d, c in per.items():
        for k, v in c.items():
            agg[kname[d]][k].append(v)
        agg[kname[d]]["_ns_" + name].append(dur[d])
    for kn, c in agg.items():
        for k, v in c.items():
            out[kn][k] = sum(v) / len(v)
        out[kn]["launches"] = len(c["_ns_" + name])
res = {}
for kn, o in out.items():
    if "FETCH_SIZE" in o:
        o["hbm_read_bytes"] = o["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in o:
        o["hbm_write_bytes"] = o["WRITE_SIZE"] * 1024
    if "hbm_read_bytes" in o and "hbm_write_bytes" in o and "_ns_fetch" in o:
        o["hbm_TBps"] = (o["hbm_read_bytes"] + o["hbm_write_bytes"]) / o["_ns_fetch"] / 1e3
    res[kn] = {k: round(v, 3) if isinstance(v, float) else v for k, v in o.items()}
json.dump(res, open(sys.argv[2], "w"), indent=1)
for kn, o in sorted(res.items(), key=lambda kv: -kv[1].get("_ns_fetch", 0) * kv[1].get("launches", 0)):
    print("%-60s x%-4d %8.1f us  read %7.1f MB  write %7.1f MB  %5.2f TB/s" % (kn[:60], o.get("launches", 0), o.get("_ns_fetch", 0) / 1e3,
          o.get("hbm_read_bytes", 0) / 1e6, o.get("hbm_write_bytes", 0) / 1e6, o.get("hbm_TBps", 0)))
