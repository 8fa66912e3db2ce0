#!/bin/bash
# dev: the fused k-NN kernel (knn_mfma.hip: knn_fused_kernel) against the multi-launch paths: bench steady state + profiled per-layer k-NN times.
# usage: scripts/dev/ab_knn_fused.sh [variant library names under lib/variants ...]
cd "${GRAFT_REPO_ROOT:-.}"
V=$PWD/livingscenes_amd/lib/variants
run() {
  timeout 600 python bench.py --cpu-instances 0 --no-fma-variant 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; pl=r['per_layer_ms_per_step']
print('$1', round(d['value']), round(d['ms_per_step'],4), 'knn0-6', [round(pl.get('knn%d'%i,0)*1e3,1) for i in range(0,7)], d['check']['handles_bit_identical'][:5], d['check']['oracle_relerr'])"
}
LS_KNN_FUSED=0 run fused=0
run fused=1
for v in "$@"; do LS_LIB_PATH=$V/$v/liblivingscenes_hip.so run $v; done
LS_KNN_FUSED=0 run fused=0
run fused=1
