#!/bin/bash
# dev: steady-state bench against the number of HIP hardware queues and steps in flight
cd "${GRAFT_REPO_ROOT:-.}"
for q in 8 16 24 32; do for f in 8 12 16 24; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-profile --cpu-instances 0 --no-fma-variant --inflight $f 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $q inflight $f', round(d['value']), round(d['ms_per_step'],4))"
done; done
