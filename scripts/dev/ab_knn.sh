#!/bin/bash
# dev: bench steady state + the k-NN builds' profiled times under each NAME=VALUE environment setting given (first: unset)
cd "${GRAFT_REPO_ROOT:-.}"
for kv in "X_UNUSED=0" "$@"; do
  env $kv python bench.py --cpu-instances 0 --no-fma-variant 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; pl=r['per_layer_ms_per_step']
print('$kv', round(d['value']), round(d['ms_per_step'],4), 'knn0-6', [round(pl.get('knn%d'%i,0)*1e3,1) for i in range(0,7)], d['check']['handles_bit_identical'][:5])"
done
