#!/bin/bash
for v in noslp; do
  echo "== variant $v"
  for r in 1 2; do LS_LIB_PATH=$PWD/scripts/diag/lib_$v.so MODE=full python scripts/diag/edge_determinism.py 2>&1 | grep -v amdgpu.ids | tail -1; done
done
