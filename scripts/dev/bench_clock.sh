#!/bin/bash
# shader clock and socket power while bench.py runs its steady state (dev tool)
cd "${GRAFT_REPO_ROOT:-.}"
python bench.py --steps 4000 --warmup 50 --no-profile --cpu-instances 0 > /tmp/bench_clock.json 2>/dev/null &
pid=$!
sleep 14
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)"; sleep 0.6; done
wait $pid
python -c "
import json; d=json.loads(open('/tmp/bench_clock.json').read().strip().splitlines()[-1]); print('value', d['value'], 'ms/step', d['ms_per_step'])"
