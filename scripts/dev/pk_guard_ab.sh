#!/bin/bash
# dev (round 6): what does the packed-fp32 guard (-fno-slp-vectorize / -fno-vectorize on edge.hip, edge_fused.hip, pointwise.hip) cost?
# Libraries built beforehand:  python scripts/dev/build_variants.py pk_edge:edge.hip=-fslp-vectorize pk_edgef:edge_fused.hip=-fslp-vectorize \
#                                  pk_pw:pointwise.hip=-fslp-vectorize,-fvectorize pk_all:edge.hip+edge_fused.hip+pointwise.hip=-fslp-vectorize,-fvectorize
#   scripts/dev/pk_guard_ab.sh       -> gpurun_out/r6/pk_guard_ab.txt
cd "${GRAFT_REPO_ROOT:-.}"
V=$PWD/livingscenes_amd/lib/variants
mkdir -p gpurun_out/r6
val() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['config']['blocks_ms'], d['check']['handles_bit_identical'][:5])"; }
one() {   # tag, library ("" = release)
  [ -n "$2" ] && export LS_LIB_PATH=$V/$2/liblivingscenes_hip.so || unset LS_LIB_PATH
  echo "== $1"
  echo -n "  steady 480 x3: "; python bench.py --no-profile --cpu-instances 0 --no-fma-variant 2>/dev/null | val
  echo -n "  driver 20 x9:  "; python bench.py --steps 20 --warmup 5 --no-profile --cpu-instances 0 --no-fma-variant 2>/dev/null | val
  echo -n "  one in flight: "; python bench.py --steps 100 --no-profile --cpu-instances 0 --no-fma-variant --inflight 1 2>/dev/null | val
  echo -n "  identity:      "; timeout 300 python scripts/dev/identity_loop.py 100 2>&1 | tail -1
}
{
one release ""
for v in pk_edge pk_edgef pk_pw pk_all; do one $v $v; done
one release-again ""
} 2>&1 | tee gpurun_out/r6/pk_guard_ab.txt
