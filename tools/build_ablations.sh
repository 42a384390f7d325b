#!/bin/bash
# builds ablation variants of libsmc_hip.so (perf experiments only; never loaded by the product)
set -e
cd "$(dirname "$0")/.."
mkdir -p particles_amd/lib/abl
# variants (each a -DSMC_<name> build of the same sources, loaded through SMC_HIP_LIBRARY):
#   BM_LEGACY        the table-free Box-Muller of rounds 1-2 (A/B of the table-driven one)
#   PHILOX_ROUNDS=7  Philox4x32-7 (Crush-resistant per Salmon et al.) instead of -10
#   TRACE            per-workgroup phase stamps (tools/trace_step.py)
#   NO_SCHEME_SPLIT  k_ancestors2 choosing systematic / stratified at run time (one kernel for both, as before)
#   NO_SU_STAGE      stratified: one Philox call per boundary instead of the tile's uniforms staged in LDS (round 4 A/B)
#   PARAMS_IN_GLOBAL k_propagate reading the model constants through the kernel argument's pointer instead of from LDS (round 4 A/B)
#   NO_BITOP3        Philox's three-way xors as two v_xor_b32 each (round 4 A/B)
for v in ${ABLS:-BM_LEGACY PHILOX_ROUNDS=7 TRACE}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=14 \
    -DABL_$v -DSMC_$v particles_amd/csrc/smc_api.hip particles_amd/csrc/smc_ops.hip particles_amd/csrc/smc_filter.hip \
    particles_amd/csrc/smc_comm.hip particles_amd/csrc/smc_sort.hip particles_amd/csrc/smc_qmc.hip -o particles_amd/lib/abl/libsmc_$v.so -ldl 2>/dev/null &
done
wait
ls -la particles_amd/lib/abl
