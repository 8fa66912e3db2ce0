#!/bin/bash
# dev (round 6): BENCH_r05's protocol (ONE warm-up step per handle, one 20-step block, 12 in flight) run N times as the first thing on a fresh box, beside
# the round-6 protocol -- does the cold allocator / single block explain the driver's 25.2k?      -> gpurun_out/r6/r5_protocol_repro.txt
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6
{
for i in 1 2 3 4 5 6; do
  LS_BENCH_R5_PROTOCOL=1 LS_BENCH_DUMP_STEPS=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-profile --cpu-instances 0 --no-fma-variant 2> gpurun_out/r6/r5rep_$i.err | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('r5-protocol run $i: value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'warmup steps', c['warmup_steps_run'], 'gap max', c['inter_completion_ms_max_per_block'])"
  grep "step completion" gpurun_out/r6/r5rep_$i.err | cut -c1-400
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-profile --cpu-instances 0 --no-fma-variant 2>/dev/null | \
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('r6-protocol: value', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'blocks_ms', c['blocks_ms'])"
} 2>&1 | tee gpurun_out/r6/r5_protocol_repro.txt
