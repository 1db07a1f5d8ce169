#!/bin/bash
# Builds an A/B variant of the library: scripts/build_variant.sh <suffix> <extra hipcc flags for mlp.hip...>
# -> uc_nerf_amd/libucnerf_hip_<suffix>.so (select it with UCNERF_LIB=...).  Run from anywhere.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SUF=$1; shift
mkdir -p /tmp/ucnerf_variant
python -m uc_nerf_amd.build >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "$@" \
      -c $R/uc_nerf_amd/csrc/mlp.hip -o /tmp/ucnerf_variant/mlp_$SUF.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/uc_nerf_amd/libucnerf_hip_$SUF.so /tmp/ucnerf_variant/mlp_$SUF.o \
      $(ls $R/uc_nerf_amd/csrc/_obj/*.o | grep -v "/mlp.o")
echo built $R/uc_nerf_amd/libucnerf_hip_$SUF.so
