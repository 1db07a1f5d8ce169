#!/bin/bash
# Bench lines and rocprofv3 kernel stats of the three routes on one box -> gpurun_out/final_* (GPU box); copy into profiles/ by hand
# (profiles/r02_bench_bf16x3_fused.json, r02_bench_f32.json, r02_bench_kernel_stats_{bf16x3_fused,bf16x3,f32}.csv)
set -e
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/final_bf16x3_fused.json 2> gpurun_out/final_bf16x3_fused.err
python bench.py --precision f32 --cpu-rays 0 --no-reuse > gpurun_out/final_f32.json 2> gpurun_out/final_f32.err
cd /tmp && export TMPDIR=/tmp
for v in bf16x3_fused bf16x3 f32; do
  rm -rf $R/gpurun_out/final_ks_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_ks_$v -- python3 $R/bench.py --cpu-rays 0 --no-reuse --precision $v > $R/gpurun_out/final_ks_$v.log 2>&1
  cp $R/gpurun_out/final_ks_$v/*/*kernel_stats.csv $R/gpurun_out/final_kernel_stats_$v.csv
done
tail -c 300 $R/gpurun_out/final_bf16x3_fused.json
