#!/bin/bash
for v in "MODE=full" "MODE=notab" "MODE=tabonly" "MODE=full LS_GEMM_BF16X3=0" "MODE=full LAYER=1" "MODE=full GPU_MAX_HW_QUEUES=4"; do
  env $v python scripts/diag/edge_determinism.py 2>&1 | grep -v amdgpu.ids | tail -4
done
