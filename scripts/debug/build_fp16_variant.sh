#!/bin/bash
# EXPERIMENT: the split kernels with fp16 operands (UCNERF_OPERAND_FP16=1, csrc/mlp_bf16.hip) -> build/variants/libucnerf_hip_fp16.so
# (all three objects of mlp_bf16.hip are rebuilt with the switch; every other object is the production build's)
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
O=/tmp/ucnerf_fp16
mkdir -p $O $R/build/variants
python -m uc_nerf_amd.build >/dev/null
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DUCNERF_OPERAND_FP16=1"
hipcc $F -c $R/uc_nerf_amd/csrc/mlp_bf16.hip -o $O/mlp_bf16.o &
hipcc $F -DUCNERF_BF16_BUILD_TERMS=1 -c $R/uc_nerf_amd/csrc/mlp_bf16.hip -o $O/mlp_bf16_plain.o &
hipcc $F -DUCNERF_BF16_BUILD_TAIL=1 -c $R/uc_nerf_amd/csrc/mlp_bf16.hip -o $O/mlp_bf16_tail.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libucnerf_hip_fp16.so $O/mlp_bf16.o $O/mlp_bf16_plain.o $O/mlp_bf16_tail.o \
      $(ls $R/uc_nerf_amd/csrc/_obj/*.o | grep -v "/mlp_bf16")
echo built $R/build/variants/libucnerf_hip_fp16.so
