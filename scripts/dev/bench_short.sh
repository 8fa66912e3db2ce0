#!/bin/bash
# dev: the bench's three regimes in one go -- steady state (480 steps, 12 in flight), the driver's protocol (20 steps), one step in flight --
# each as "value ms_per_step" (no profiled pass, no CPU leg).   scripts/dev/bench_short.sh [label]
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],4), d['check']['handles_bit_identical'][:5])"; }
python bench.py --no-profile --cpu-instances 0 --no-fma-variant 2>/dev/null | line "${1:-x} steady480"
python bench.py --steps 20 --warmup 5 --no-profile --cpu-instances 0 --no-fma-variant 2>/dev/null | line "${1:-x} driver20"
python bench.py --no-profile --cpu-instances 0 --no-fma-variant --inflight 1 2>/dev/null | line "${1:-x} one-in-flight"
