#!/bin/bash
# dev: marginal cost of the 32-point attention layers (edge_ft_* kernels + operand images) in the 12-in-flight bench
cd "${GRAFT_REPO_ROOT:-.}"
python scripts/dev/build_variants.py devknobs:model.hip=-DLS_DEV_KNOBS > /dev/null || exit 1
export LS_LIB_PATH=$PWD/livingscenes_amd/lib/variants/devknobs/liblivingscenes_hip.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), d['ms_per_step'])"; }
for fam in none hi32 attn none hi32; do
  LS_SKIP=$fam python bench.py --cpu-instances 0 --no-profile --no-fma-variant 2>/dev/null | tail -1 | line $fam
done
