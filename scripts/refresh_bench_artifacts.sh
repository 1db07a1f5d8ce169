#!/bin/bash
# Bench lines, rocprofv3 kernel stats and PMC passes of one box -> gpurun_out/final_* (GPU box); scripts/collect_profiles.sh copies them into profiles/rNN/
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
# training: the library's fused pass (1024 x 128) and the rendering() drop-in step (2000 x 90), per-kernel times
rm -rf $R/gpurun_out/final_ks_train $R/gpurun_out/final_ks_dropin_train
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_ks_train -- python3 $R/scripts/time_train_step.py > $R/gpurun_out/final_ks_train.log 2>&1
cp $R/gpurun_out/final_ks_train/*/*kernel_stats.csv $R/gpurun_out/final_kernel_stats_train_step.csv
MODE=train rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final_ks_dropin_train -- python3 $R/scripts/time_dropin.py > $R/gpurun_out/final_ks_dropin_train.log 2>&1
cp $R/gpurun_out/final_ks_dropin_train/*/*kernel_stats.csv $R/gpurun_out/final_kernel_stats_dropin_train.csv
cd $R
# PMC passes (separate runs per counter group): the fused forward kernel of the bench, and the training-step kernels
PREC=bf16x3_fused KERNEL=mlp_fwd_bf16 bash scripts/pmc_mlp.sh final_pmc_fused > gpurun_out/final_pmc_fused.log 2>&1
python scripts/pmc_summary.py gpurun_out/final_pmc_fused gpurun_out/final_mlp_bf16_fused_hbm_traffic.json > gpurun_out/final_pmc_fused_summary.log 2>&1 || true
KERNEL="mlp_bwd_chain|mlp_wgrad|mlp_fwd_kernel|feat_gather" SCRIPT=scripts/time_train_step.py bash scripts/pmc_mlp.sh final_pmc_train > gpurun_out/final_pmc_train.log 2>&1
python scripts/pmc_by_kernel.py gpurun_out/final_pmc_train gpurun_out/final_train_step_hbm_traffic.json > gpurun_out/final_pmc_train_summary.log 2>&1 || true
# round 4: the strong-scaling shard's step (512 rays) as a kernel trace with the gaps between launches, and the rendering() training step at 2000 / 250 rays
bash scripts/prof_strong.sh final > gpurun_out/final_strong512.log 2>&1 || true
bash scripts/prof_dropin_train.sh final > gpurun_out/final_dropin_train_trace.log 2>&1 || true
# the data-parallel step through a ONE-rank RCCL group (the collective's code path on one GPU)
UCNERF_BENCH_GROUP_AT_1=1 python bench.py --cpu-rays 0 2> gpurun_out/final_rccl1.err | grep '^{' > gpurun_out/final_rccl1.json || true
# the self-launched four-rank line (all ranks on this GPU, gloo for the collective; six processes at most may share the card): only the JSON line is kept
UCNERF_BENCH_BACKEND=gloo python bench.py --gpus 4 --steps 20 --warmup 5 2> gpurun_out/final_n4_gloo.err | grep '^{' > gpurun_out/final_n4_gloo.json || true
tail -c 300 $R/gpurun_out/final_bf16x3_fused.json
