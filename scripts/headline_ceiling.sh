#!/bin/bash
# One box, one call: the four-step ceiling of the split-bf16 formulation (scripts/micro/headline_ceiling.hip), then the
# headline kernel itself under the same power / clock probe, then the bench line that carries the kernel's HIP-event rate.
#   gpurun -- bash scripts/headline_ceiling.sh        -> gpurun_out/ceiling/{micro.jsonl,kernel_fused.json,kernel_two.json,bench.json}
set -e
out=gpurun_out/ceiling
mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -w -o $out/headline_ceiling scripts/micro/headline_ceiling.hip
$out/headline_ceiling 2.5 > $out/micro.jsonl
echo "micro done"; cat $out/micro.jsonl
python scripts/power_probe.py --seconds 4 --tag kernel_bf16x3_fused > $out/kernel_fused.json
python scripts/power_probe.py --seconds 4 --precision bf16x3 --tag kernel_bf16x3_two_kernel > $out/kernel_two.json
python bench.py --steps 400 --headline-only --cpu-rays 0 > $out/bench.json
$out/headline_ceiling 1.5 > $out/micro_after.jsonl
python - <<'PY'
import json
m = [json.loads(l) for l in open("gpurun_out/ceiling/micro.jsonl")]
b = json.loads(open("gpurun_out/ceiling/bench.json").read().strip().splitlines()[-1])
k = json.loads(open("gpurun_out/ceiling/kernel_fused.json").read().strip().splitlines()[-1])
best = {}
for r in m:
    best[r["config"]] = max(best.get(r["config"], 0), r["executed_tflops"])
print(json.dumps({"micro_best_executed_tflops": best, "kernel_roofline": b["roofline"], "kernel_power_w": k.get("power_w_mean"), "kernel_sclk": k.get("sclk_mhz_mean")}))
PY
