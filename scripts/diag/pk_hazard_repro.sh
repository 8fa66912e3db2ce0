#!/bin/bash
# Reproducibility defect of DESIGN 4.3: the reproducer (pk_hazard_repro.py) under the release library and under libraries whose edge.hip is built WITH
# hipcc's SLP vectoriser (compiler-formed v_pk_*_f32), plain and with n + 1 wait states in front of every DPP instruction of the reductions (LS_DPP_NOP=n).
#   scripts/diag/pk_hazard_repro.sh          (GPU box; writes gpurun_out/pk_hazard.txt)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
python scripts/dev/build_variants.py pk:edge.hip=-fslp-vectorize pk_nop1:edge.hip=-fslp-vectorize,-DLS_DPP_NOP=1 pk_nop7:edge.hip=-fslp-vectorize,-DLS_DPP_NOP=7 \
    nop7:edge.hip=-DLS_DPP_NOP=7 > /dev/null || exit 1
V=$PWD/livingscenes_amd/lib/variants
{
for mode in bf16x3 h2; do
  [ $mode = h2 ] && unset LS_GEMM_MODE || export LS_GEMM_MODE=$mode
  timeout 300 python scripts/diag/pk_hazard_repro.py 12 2>&1 | tail -1
  for v in pk pk_nop1 pk_nop7 nop7; do LS_LIB_PATH=$V/$v/liblivingscenes_hip.so timeout 300 python scripts/diag/pk_hazard_repro.py 12 2>&1 | tail -1; done
done
} | tee gpurun_out/pk_hazard.txt
