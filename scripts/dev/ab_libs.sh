#!/bin/bash
# dev: bench steady state + profiled per-layer k-NN / attention / table times under the default library and each variant library named
# (scripts/dev/build_variants.py NAME:file.hip=-DFLAG ...).   usage: scripts/dev/ab_libs.sh [variant ...]
cd "${GRAFT_REPO_ROOT:-.}"
V=$PWD/livingscenes_amd/lib/variants
run() {
  timeout 600 python bench.py --cpu-instances 0 --no-fma-variant 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; pl=r['per_layer_ms_per_step']
f=lambda k: [round(pl.get(k+str(i),0)*1e3,1) for i in range(0,7)]
print('$1', round(d['value']), round(d['ms_per_step'],4), 'knn', f('knn'), 'attn', f('edge_attn')[2:], 'glob', f('gemm_glob')[2:], d['check']['handles_bit_identical'][:5])"
}
run default
for v in "$@"; do LS_LIB_PATH=$V/$v/liblivingscenes_hip.so run $v; done
run default
