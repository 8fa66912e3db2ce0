#!/bin/bash
# dev: bench steady state + the k-NN builds' profiled times under each variant library given (names under lib/variants; first: the product)
cd "${GRAFT_REPO_ROOT:-.}"
V=livingscenes_amd/lib/variants
for name in default "$@"; do
  if [ $name = default ]; then unset LS_LIB_PATH; else export LS_LIB_PATH=$PWD/$V/$name/liblivingscenes_hip.so; fi
  python bench.py --cpu-instances 0 --no-fma-variant 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; pl=r['per_layer_ms_per_step']
print('$name', round(d['value']), round(d['ms_per_step'],4), 'knn0-6', [round(pl.get('knn%d'%i,0)*1e3,1) for i in range(0,7)], d['check']['handles_bit_identical'][:5])"
done
