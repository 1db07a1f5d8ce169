"""Per (kernel, grid size) averages of the passes written by scripts/pmc_mvs.sh -> JSON + a table: instruction counts, busy fractions, HBM bytes
(FETCH_SIZE x 2 on gfx950: 64 B tallied per 128-B request; WRITE_SIZE in KB), L2 hit rate, duration."""
import collections
import csv
import glob
import json
import sys

base = sys.argv[1]
out = collections.defaultdict(dict)
for run in sorted(glob.glob(base + "/*/")):
    name = run.rstrip("/").split("/")[-1]
    fs = sorted(glob.glob(run + "/*/*counter_collection.csv"))
    if not fs:
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    meta = {}
    for r in csv.DictReader(open(fs[-1])):
        d = r["Dispatch_Id"]
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        meta[d] = (r["Kernel_Name"].split("(")[0].replace("ucnerf::", ""), int(r.get("Grid_Size", 0) or 0), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d, c in per.items():
        key = "%s grid=%d" % meta[d][:2]
        for k, v in c.items():
            agg[key][k].append(v)
        agg[key]["_ns_" + name].append(meta[d][2])
    for key, c in agg.items():
        for k, v in c.items():
            v = sorted(v)[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)] or v          # middle half: the first launches run cold
            out[key][k] = sum(v) / len(v)
res = {}
for key, o in out.items():
    if "FETCH_SIZE" in o:
        o["hbm_read_bytes"] = o["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in o:
        o["hbm_write_bytes"] = o["WRITE_SIZE"] * 1024
    ns = o.get("_ns_grbm") or o.get("_ns_fetch")
    if ns and "hbm_read_bytes" in o and "hbm_write_bytes" in o:
        o["hbm_TBps"] = (o["hbm_read_bytes"] + o["hbm_write_bytes"]) / ns / 1e3
    if "TCC_HIT_sum" in o:
        o["l2_hit_rate"] = o["TCC_HIT_sum"] / max(1.0, o["TCC_HIT_sum"] + o["TCC_MISS_sum"])
    if "SQ_BUSY_CYCLES" in o and "SQ_ACTIVE_INST_VALU" in o:
        # SQ_BUSY_CYCLES counts per SE (x32 on this part: see pmc_summary.py); busy fractions are better taken against wave cycles
        o["valu_active_per_wave_cycle"] = o["SQ_ACTIVE_INST_VALU"] / max(1.0, o["SQ_WAVE_CYCLES"])
        o["vmem_active_per_wave_cycle"] = o["SQ_ACTIVE_INST_VMEM"] / max(1.0, o["SQ_WAVE_CYCLES"])
        o["wait_inst_per_wave_cycle"] = o["SQ_WAIT_INST_ANY"] / max(1.0, o["SQ_WAVE_CYCLES"])
    if "SQ_WAVES" in o and o["SQ_WAVES"]:
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"):
            if k in o:
                o[k.lower().replace("sq_", "") + "_per_wave"] = o[k] / o["SQ_WAVES"]
    if "TA_TA_BUSY_sum" in o and "GRBM_GUI_ACTIVE" in o:
        o["ta_busy_frac_of_256_cus"] = o["TA_TA_BUSY_sum"] / 256.0 / max(1.0, o["GRBM_GUI_ACTIVE"])
    res[key] = {k: round(v, 4) if isinstance(v, float) else v for k, v in o.items()}
json.dump(res, open(sys.argv[2], "w"), indent=1)
for key, o in sorted(res.items()):
    print("%-52s %7.1f us  rd %6.1f MB wr %6.1f MB %5.2f TB/s  L2 hit %.2f  valu/wave %6.0f vmem_rd/wave %5.0f  valu-act %.2f vmem-act %.2f wait %.2f  TA busy %.2f" % (
        key[:52], (o.get("_ns_grbm") or o.get("_ns_fetch") or 0) / 1e3, o.get("hbm_read_bytes", 0) / 1e6, o.get("hbm_write_bytes", 0) / 1e6, o.get("hbm_TBps", 0),
        o.get("l2_hit_rate", 0), o.get("insts_valu_per_wave", 0), o.get("insts_vmem_rd_per_wave", 0), o.get("valu_active_per_wave_cycle", 0),
        o.get("vmem_active_per_wave_cycle", 0), o.get("wait_inst_per_wave_cycle", 0), o.get("ta_busy_frac_of_256_cus", 0)))
