#!/bin/bash
# Copies what scripts/refresh_bench_artifacts.sh (and the test run) left in gpurun_out/ into profiles/ under this round's prefix:
#   scripts/collect_profiles.sh r03
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$1
G=$R/gpurun_out
cp $G/final_bf16x3_fused.json $R/profiles/${P}_bench_bf16x3_fused.json
cp $G/final_f32.json $R/profiles/${P}_bench_f32.json
for v in bf16x3_fused bf16x3 f32; do cp $G/final_kernel_stats_$v.csv $R/profiles/${P}_bench_kernel_stats_$v.csv; done
cp $G/final_kernel_stats_train_step.csv $R/profiles/${P}_train_step_kernel_stats.csv
cp $G/final_kernel_stats_dropin_train.csv $R/profiles/${P}_dropin_train_kernel_stats.csv
cp $G/final_mlp_bf16_fused_hbm_traffic.json $R/profiles/${P}_mlp_bf16_fused_hbm_traffic.json
cp $G/final_train_step_hbm_traffic.json $R/profiles/${P}_train_step_hbm_traffic.json
python3 - "$G/final_pmc_train" "$R/profiles/${P}_train_step_sq_counters.json" <<'PY'
import collections, csv, glob, json, sys
out = collections.defaultdict(dict)
for name in ("sq1", "sq2", "grbm", "tcc"):
    fs = sorted(glob.glob(sys.argv[1] + "/" + name + "/*/*counter_collection.csv"))
    if not fs:
        continue
    per, kn, dur = collections.defaultdict(lambda: collections.defaultdict(float)), {}, {}
    for r in csv.DictReader(open(fs[-1])):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        kn[r["Dispatch_Id"]] = r["Kernel_Name"].split("(")[0]
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d, c in per.items():
        for k, v in c.items():
            agg[kn[d]][k].append(v)
        agg[kn[d]]["_duration_ns_" + name].append(dur[d])
    for k, c in agg.items():
        for a, b in c.items():
            out[k][a] = sum(b) / len(b)
for k, o in out.items():
    if "SQ_WAVE_CYCLES" in o:
        w = o["SQ_WAVE_CYCLES"]
        o["frac_wait_any"] = o.get("SQ_WAIT_ANY", 0) / w
        o["frac_wait_inst_any"] = o.get("SQ_WAIT_INST_ANY", 0) / w
        o["frac_active_inst_any"] = o.get("SQ_ACTIVE_INST_ANY", 0) / w
    if "GRBM_GUI_ACTIVE" in o:
        # GRBM_GUI_ACTIVE counts cycles of the whole dispatch window (queue pick-up, wave launch and drain included), so cycles / duration is a
        # clock only for long launches: below 50 us it over-reads (5.4 "GHz" on a 3-us kernel) and is reported as invalid instead
        if o["_duration_ns_grbm"] >= 50e3:
            o["clock_ghz"] = o["GRBM_GUI_ACTIVE"] / 8 / o["_duration_ns_grbm"]
        else:
            o["clock_ghz"] = None
            o["clock_ghz_note"] = "launch under 50 us: GRBM_GUI_ACTIVE / duration is not a clock"
    if "TCC_HIT_sum" in o:
        o["l2_hit_rate"] = o["TCC_HIT_sum"] / (o["TCC_HIT_sum"] + o["TCC_MISS_sum"])
json.dump({"source": "scripts/pmc_mlp.sh (rocprofv3 --pmc, separate passes) on scripts/time_train_step.py (1024 rays x 128 samples)", "kernels": out},
          open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
[ -s $G/final_n2_gloo.json ] && cp $G/final_n2_gloo.json $R/profiles/${P}_bench_n2_gloo_rehearsal.json
[ -s $G/final_n4_gloo.json ] && cp $G/final_n4_gloo.json $R/profiles/${P}_bench_n4_gloo_rehearsal.json
[ -s $G/final_rccl1.json ] && cp $G/final_rccl1.json $R/profiles/${P}_bench_rccl_one_rank_group.json
for v in repack hoisted; do
  [ -s $G/strong_final_$v.kernel_stats.csv ] && cp $G/strong_final_$v.kernel_stats.csv $R/profiles/${P}_strong512_kernel_stats_$v.csv
  [ -s $G/strong_final_$v.gaps.json ] && cp $G/strong_final_$v.gaps.json $R/profiles/${P}_strong512_launch_gaps_$v.json
done
for n in 2000 250; do
  [ -s $G/dropin_train_final_$n.kernel_stats.csv ] && cp $G/dropin_train_final_$n.kernel_stats.csv $R/profiles/${P}_dropin_train_kernel_stats_$n.csv
  [ -s $G/dropin_train_final_$n.gaps.json ] && cp $G/dropin_train_final_$n.gaps.json $R/profiles/${P}_dropin_train_launch_gaps_$n.json
done
[ -f $G/parity_errors.json ] && cp $G/parity_errors.json $R/profiles/${P}_parity_errors.json
echo collected into profiles/${P}_*
