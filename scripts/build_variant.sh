#!/bin/bash
# scripts/build_variant.sh NAME [-DFLAG ...]: the library with extra compiler flags -> chimera_amd/lib/variants/libchimera_hip_NAME.so
N=$1; shift
mkdir -p chimera_amd/lib/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -std=c++17 -Wall -Wno-unused-function -Wno-unused-variable -Wno-invalid-offsetof -cuid=chimera_hip "$@" \
  -o chimera_amd/lib/variants/libchimera_hip_$N.so chimera_amd/csrc/chimera_hip.hip -L/opt/rocm/lib -lrccl
