#!/bin/bash
for v in "" ldsnoslp online4; do
  echo "== variant ${v:-current}"
  if [ -n "$v" ]; then export LS_LIB_PATH=$PWD/scripts/diag/lib_$v.so; else unset LS_LIB_PATH; fi
  MODE=full python scripts/diag/edge_determinism.py 2>&1 | grep -v amdgpu.ids | tail -1
  python bench.py --no-fma-variant --cpu-instances 0 --steps 240 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], {k:v for k,v in d['roofline']['per_layer_ms_per_step'].items() if 'edge_attn' in k}, d['roofline']['breakdown_ms_per_step']['edge_attn'])"
done
