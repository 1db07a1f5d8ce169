#!/bin/bash
# Builds an A/B variant of the library: [SRC=mlp_bf16] scripts/build_variant.sh <suffix> <extra hipcc flags for $SRC.hip...>
# -> build/variants/libucnerf_hip_<suffix>.so (select it with UCNERF_LIB=...; the A/B scripts look there).  Run from anywhere.
# Variants never land inside the package directory: only the production library is importable from uc_nerf_amd/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
SUF=$1; shift
SRC=${SRC:-mlp}
OUT=$R/build/variants
mkdir -p /tmp/ucnerf_variant $OUT
python -m uc_nerf_amd.build >/dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "$@" \
      -c $R/uc_nerf_amd/csrc/$SRC.hip -o /tmp/ucnerf_variant/${SRC}_$SUF.o
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libucnerf_hip_$SUF.so /tmp/ucnerf_variant/${SRC}_$SUF.o \
      $(ls $R/uc_nerf_amd/csrc/_obj/*.o | grep -v "/$SRC.o")
echo built $OUT/libucnerf_hip_$SUF.so
