#!/bin/bash
# dev: the LDS-staged attention kernel (edge_staged.hip) against the row-gather kernel: bench steady state + profiled per-layer launch times of the
# attention layers and their table GEMMs under LS_EDGE_STAGED=0 | 1.   scripts/dev/ab_staged.sh [extra bench args]
cd "${GRAFT_REPO_ROOT:-.}"
for mode in 0 1 0 1; do
  LS_EDGE_STAGED=$mode timeout 600 python bench.py --cpu-instances 0 --no-fma-variant "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; pl=r['per_layer_ms_per_step']
print('staged=$mode', round(d['value']), round(d['ms_per_step'],4), 'attn2-6', [round(pl.get('edge_attn%d'%i,0)*1e3,1) for i in range(2,7)], 'tables1-4', [round(pl.get('gemm_edge%d'%i,0)*1e3,1) for i in range(1,5)], 'pool', round(pl.get('edge_pool1',0)*1e3,1), d['check']['handles_bit_identical'][:5])"
done
