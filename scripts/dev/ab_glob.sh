#!/bin/bash
# dev: bench steady state + the global-conv family's profiled launch times (mean + GEMM per layer)
cd "${GRAFT_REPO_ROOT:-.}"
for kv in "X_UNUSED=0" "$@"; do
  env $kv python bench.py --cpu-instances 0 --no-fma-variant 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; pl=r['per_layer_ms_per_step']
print('$kv', round(d['value']), round(d['ms_per_step'],4), 'mean2-6', [round(pl.get('mean%d'%i,0)*1e3,1) for i in range(2,7)], 'gemm_glob', [round(pl.get('gemm_glob%d'%i,0)*1e3,1) for i in range(2,7)], 'tail', round(pl.get('tail0',0)*1e3,1), round(pl.get('gemm_tail0',0)*1e3,1))"
done
