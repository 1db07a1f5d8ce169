#!/bin/bash
# Builds an A/B variant of the library: [SRC=mlp_bf16] scripts/build_variant.sh <suffix> <extra hipcc flags for $SRC.hip...>
# -> uc_nerf_amd/libucnerf_hip_<suffix>.so (select it with UCNERF_LIB=...).  Run from anywhere.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SUF=$1; shift
SRC=${SRC:-mlp}
mkdir -p /tmp/ucnerf_variant
python -m uc_nerf_amd.build >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "$@" \
      -c $R/uc_nerf_amd/csrc/$SRC.hip -o /tmp/ucnerf_variant/${SRC}_$SUF.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/uc_nerf_amd/libucnerf_hip_$SUF.so /tmp/ucnerf_variant/${SRC}_$SUF.o \
      $(ls $R/uc_nerf_amd/csrc/_obj/*.o | grep -v "/$SRC.o")
echo built $R/uc_nerf_amd/libucnerf_hip_$SUF.so
