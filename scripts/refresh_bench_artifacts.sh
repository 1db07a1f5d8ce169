#!/bin/bash
# Bench lines and rocprofv3 kernel stats of both precisions on one box -> gpurun_out/final_* (GPU box); copy into profiles/ by hand
# (profiles/r02_bench_bf16x3.json, r02_bench_f32.json, r02_bench_kernel_stats_{bf16x3,f32}.csv)
set -e
R=$GRAFT_REPO_ROOT
cd $R
python bench.py > gpurun_out/final_bf16x3.json 2> gpurun_out/final_bf16x3.err
python bench.py --precision f32 --cpu-rays 0 --no-reuse > gpurun_out/final_f32.json 2> gpurun_out/final_f32.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/final_ks_bf16x3 $R/gpurun_out/final_ks_f32
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_ks_bf16x3 -- python3 $R/bench.py --cpu-rays 0 --no-reuse > $R/gpurun_out/final_ks_bf16x3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_ks_f32 -- python3 $R/bench.py --cpu-rays 0 --no-reuse --precision f32 > $R/gpurun_out/final_ks_f32.log 2>&1
cp $R/gpurun_out/final_ks_bf16x3/*/*kernel_stats.csv $R/gpurun_out/final_kernel_stats_bf16x3.csv
cp $R/gpurun_out/final_ks_f32/*/*kernel_stats.csv $R/gpurun_out/final_kernel_stats_f32.csv
tail -c 300 $R/gpurun_out/final_bf16x3.json
