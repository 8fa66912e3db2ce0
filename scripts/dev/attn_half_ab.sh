#!/bin/bash
# dev (round 6): upper bound of a half-width attention table -- the attention kernel of layers 2 - 4 alone (scripts/dev/attn_layers.py) and the
# bench under the release library and under fq_half (edge.hip built with -DLS_FQ_HALF_GATHER: the `dir` gathers dropped, WRONG results).
#   python scripts/dev/build_variants.py fq_half:edge.hip=-DLS_FQ_HALF_GATHER;  scripts/dev/attn_half_ab.sh   -> gpurun_out/r6/attn_halfwidth_ab.txt
cd "${GRAFT_REPO_ROOT:-.}"
V=$PWD/livingscenes_amd/lib/variants
mkdir -p gpurun_out/r6
val() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
{
echo "== release: attention kernel alone (us per launch, B = 64)"; python scripts/dev/attn_layers.py --modes 0 --reps 9 2>&1 | grep "^mode"
echo "== fq_half (dir gathers dropped): attention kernel alone"; LS_LIB_PATH=$V/fq_half/liblivingscenes_hip.so python scripts/dev/attn_layers.py --modes 0 --reps 9 2>&1 | grep "^mode"
for rep in 1 2; do
echo -n "release steady 480 x3: "; python bench.py --no-profile --cpu-instances 0 --no-fma-variant 2>/dev/null | val
echo -n "fq_half steady 480 x3: "; LS_LIB_PATH=$V/fq_half/liblivingscenes_hip.so python bench.py --no-profile --cpu-instances 0 --no-fma-variant 2>/dev/null | val
done
} 2>&1 | tee gpurun_out/r6/attn_halfwidth_ab.txt
