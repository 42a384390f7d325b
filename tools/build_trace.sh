#!/bin/bash
set -e
cd "$(dirname "$0")/.."
mkdir -p particles_amd/lib/abl
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -DSMC_TRACE \
  particles_amd/csrc/smc_api.hip particles_amd/csrc/smc_ops.hip particles_amd/csrc/smc_filter.hip particles_amd/csrc/smc_comm.hip \
  -o particles_amd/lib/abl/libsmc_TRACE.so -ldl 2>/dev/null
ls -la particles_amd/lib/abl/libsmc_TRACE.so
