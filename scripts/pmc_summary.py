"""Summarises the CSVs written by scripts/pmc_mlp.sh into per-launch averages (coarse / fine dispatches)."""
import collections
import csv
import glob
import os
import json
import sys

base = sys.argv[1]
out = {}
for name in ("sq1", "sq2", "fetch", "write", "grbm", "tcc"):
    fs = glob.glob(base + "/" + name + "/*/*counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):        # (the newest pass when earlier ones share the directory)
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ds = sorted(agg, key=int)
    for kind, sel in (("coarse", ds[2::2]), ("fine", ds[3::2])):      # dispatches alternate; skip the warm-up pair
        for c in agg[sel[0]]:
            out.setdefault(kind, {})[c] = sum(agg[d][c] for d in sel) / len(sel)
        out[kind]["_duration_ns_" + name] = sum(dur[d] for d in sel) / len(sel)
for kind, o in out.items():
    samples = 262144 if kind == "coarse" else 786432
    if "FETCH_SIZE" in o:
        o["hbm_read_bytes_corrected"] = o["FETCH_SIZE"] * 1024 * 2     # gfx950: FETCH_SIZE tallies 64 B per 128-B request
        o["hbm_write_bytes"] = o["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in o:
        o["l2_hit_rate"] = o["TCC_HIT_sum"] / (o["TCC_HIT_sum"] + o["TCC_MISS_sum"])
    for c in ("VALU", "SALU", "LDS", "MFMA", "VMEM_RD"):
        if "SQ_INSTS_" + c in o:
            o[c.lower() + "_insts_per_tile"] = o["SQ_INSTS_" + c] / (samples / 32)
    if "GRBM_GUI_ACTIVE" in o:
        # (cycles of the whole dispatch window / duration: a clock only for long launches -- under 50 us it over-reads and is dropped)
        o["clock_ghz"] = o["GRBM_GUI_ACTIVE"] / 8 / o["_duration_ns_grbm"] if o["_duration_ns_grbm"] >= 50e3 else None
        if "SQ_VALU_MFMA_BUSY_CYCLES" in o and o["clock_ghz"]:
            o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (o["_duration_ns_sq1"] * o["clock_ghz"])
res = {"source": "scripts/pmc_mlp.sh (rocprofv3 --pmc, separate passes) on python3 bench.py --steps 4 --warmup 1 --cpu-rays 0",
       "coarse": out.get("coarse"), "fine": out.get("fine")}
if out.get("coarse", {}).get("hbm_read_bytes_corrected"):
    res["bytes_per_launch"] = sum(out[k]["hbm_read_bytes_corrected"] + out[k]["hbm_write_bytes"] for k in ("coarse", "fine")) / 2
json.dump(res, open(sys.argv[2], "w"), indent=1)
for k in ("coarse", "fine"):
    o = out[k]
    print(k, {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in o.items() if kk in
              ("valu_insts_per_tile", "salu_insts_per_tile", "lds_insts_per_tile", "mfma_insts_per_tile", "vmem_rd_insts_per_tile", "clock_ghz", "mfma_busy_frac", "l2_hit_rate", "hbm_read_bytes_corrected", "hbm_write_bytes", "_duration_ns_sq1")})
