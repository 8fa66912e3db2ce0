#!/bin/bash
# dev (round 6): how the in-flight steps share the hardware queues -- FPS chain on the caller's stream (LS_FPS_SIDE=0: one stream per step) against the side
# stream, by in-flight depth and number of hardware queues.  Steady state (480 steps x 3 blocks) and the 20-step protocol.   -> gpurun_out/r6/streams_ab.txt
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r6
val() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"; }
run() {  # label, env..., -- bench args
  label=$1; shift
  echo -n "$label steady: "; env "$@" python bench.py --no-profile --cpu-instances 0 --no-fma-variant $ARGS 2>/dev/null | val
  echo -n "$label 20-step: "; env "$@" python bench.py --steps 20 --warmup 5 --no-profile --cpu-instances 0 --no-fma-variant $ARGS 2>/dev/null | val
}
{
for d in 12 16; do
  ARGS="--inflight $d"
  run "side-stream FPS, depth $d, 16 queues" X=1
  run "caller-stream FPS, depth $d, 16 queues" LS_FPS_SIDE=0
done
ARGS="--inflight 16"; run "caller-stream FPS, depth 16, 24 queues" LS_FPS_SIDE=0 GPU_MAX_HW_QUEUES=24
ARGS="--inflight 20"; run "caller-stream FPS, depth 20, 24 queues" LS_FPS_SIDE=0 GPU_MAX_HW_QUEUES=24
ARGS="--inflight 12"; run "caller-stream FPS, depth 12, 12 queues" LS_FPS_SIDE=0 GPU_MAX_HW_QUEUES=12
ARGS="--inflight 8"; run "caller-stream FPS, depth 8, 8 queues" LS_FPS_SIDE=0 GPU_MAX_HW_QUEUES=8
ARGS="--inflight 12"; run "side-stream FPS, depth 12, 16 queues (again)" X=1
} 2>&1 | tee gpurun_out/r6/streams_ab.txt
