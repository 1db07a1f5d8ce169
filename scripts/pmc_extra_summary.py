"""Summarises scripts/pmc_extra.sh output (coexec + lds passes) into per-fine-launch averages + derived fractions.
usage: python scripts/pmc_extra_summary.py gpurun_out/<dir> out.json [clock_ghz]"""
import collections
import csv
import glob
import json
import sys

base, out_path = sys.argv[1], sys.argv[2]
clock = float(sys.argv[3]) if len(sys.argv) > 3 else None
res = {"source": "scripts/pmc_extra.sh (rocprofv3 --pmc, two passes) on python3 bench.py --steps 4 --warmup 1 --cpu-rays 0 --no-reuse; "
                 "fine-pass launches of mlp_fwd_bf16_kernel, averages per launch, summed over the chip's 1024 SIMDs"}
for name in ("coexec", "lds"):
    fs = glob.glob(base + "/" + name + "/*/*counter_collection.csv")
    if not fs:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = {}
    for r in csv.DictReader(open(fs[0])):
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ds = sorted(agg, key=int)
    longest = max(dur[d] for d in ds)
    fine = [d for d in ds if dur[d] > 0.6 * longest]                 # the fine-pass launches (3x the samples of the coarse ones)
    o = {c: sum(agg[d][c] for d in fine) / len(fine) for c in agg[fine[0]]}
    o["_duration_ns"] = sum(dur[d] for d in fine) / len(fine)
    o["_launches"] = len(fine)
    res[name] = o
d = {}
if "coexec" in res and clock:
    c = res["coexec"]
    cyc = c["_duration_ns"] * clock * 1024                           # SIMD-cycles of the launch
    d["launch_cycles_x_simds"] = cyc
    d["clock_ghz_assumed"] = clock
    d["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc
    d["valu_active_frac (x4: counted per quad-cycle)"] = 4 * c["SQ_ACTIVE_INST_VALU"] / cyc
    d["valu_under_mfma_frac_of_valu"] = c["SQ_VALU_MFMA_COEXEC_CYCLES"] / (4 * c["SQ_ACTIVE_INST_VALU"])
if "lds" in res:
    l = res["lds"]
    d["lds_bank_conflict_frac"] = l["SQ_LDS_BANK_CONFLICT"] / l["SQ_LDS_IDX_ACTIVE"]
res["derived"] = d
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(d, indent=1))
