#!/bin/bash
# Upper bound of what fusing the gather into the MLP kernel could save: the headline step with the feature
# round trip REMOVED by timing switches -- gather without its feature stores (UCNERF_GATHER_EXP=1), MLP without its feature /
# point loads (UCNERF_BF16_EXP=1024); results are wrong, timings valid -- at 4096 and at 512 rays per GPU, A/B interleaved.
# Needs uc_nerf_amd/libucnerf_hip_f1bound.so (see DESIGN.md section 4.8 for the build line).
R=$(cd "$(dirname "$0")/.." && pwd)
for rays in 4096 512; do
  steps=200; [ $rays = 512 ] && steps=800
  for rep in 1 2; do
    for v in - f1bound; do
      if [ "$v" = "-" ]; then unset UCNERF_LIB; else export UCNERF_LIB=$R/build/variants/libucnerf_hip_$v.so; fi
      timeout -k 10 120 python $R/bench.py --rays $rays --steps $steps --cpu-rays 0 --no-reuse 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print(json.dumps({'variant': '$v', 'rays': $rays, 'ms_per_step': d['ms_per_step'], 'mlp_launch_ms': r['avg_launch_ms'], 'mlp_share': d['mlp_share_of_step']}))" || exit 1
    done
  done
done
