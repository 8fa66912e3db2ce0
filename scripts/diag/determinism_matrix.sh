#!/bin/bash
# run scripts/diag/determinism.py under a matrix of A/B switches; print one summary line per configuration
run() { echo "=== $*"; env "$@" python scripts/diag/determinism.py 8 2>&1 | grep -v amdgpu.ids | awk '/IDENTICAL/{a++} /\[\(/{b++} END{print "identical:",a," differing:",b}'; }
run X=1
run LS_FPS_SIDE=0
run LS_GEMM_BF16X3=0
run LS_KNN_HINTS=auto
run LS_KNN_HINTS=prev
run LS_KNN_AUTOHINTS=0
run LS_KNN_FILTER=0
run LS_KNN_SEEDS=0
run GPU_MAX_HW_QUEUES=4
