#!/bin/bash
# times the wide GEMM under every variant library given as argument (names under livingscenes_amd/lib/variants), plus the product library
cd "${GRAFT_REPO_ROOT:-.}"
LS_TAG="product pp=1" LS_GEMM_W2_PP=1 python scripts/dev/w2_time.py 2>/dev/null
LS_TAG="product pp=0" LS_GEMM_W2_PP=0 python scripts/dev/w2_time.py 2>/dev/null
LS_TAG="product pp=0 W planes LDS-direct" LS_GEMM_W2_DIRECT=1 python scripts/dev/w2_time.py 2>/dev/null
for v in "$@"; do
  LS_TAG="$v" LS_LIB_PATH=livingscenes_amd/lib/variants/$v/liblivingscenes_hip.so python scripts/dev/w2_time.py 2>/dev/null
done
