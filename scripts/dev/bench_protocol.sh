#!/bin/bash
# dev (round 6): the driver's command and the in-flight depth on the 20-step protocol, block statistics per run.
#   scripts/dev/bench_protocol.sh <tag> [depths...]      -> gpurun_out/r6/<tag>_*.json + one summary line per run
cd "${GRAFT_REPO_ROOT:-.}"
tag=${1:-p}; shift
depths=${@:-"8 12"}
mkdir -p gpurun_out/r6
sum() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); c = d["config"]
print(sys.argv[1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "inflight", c["steps_in_flight"], "blocks_ms", c["blocks_ms"],
      "max/min", c["block_max_over_min"], "gapmax", c["inter_completion_ms_max_per_block"], "one-in-flight ms", c["ms_per_step_one_in_flight_unprofiled"],
      "clk", c["gpu_clock_power_idle"], c["gpu_clock_power_after_warmup"], c["gpu_clock_power_after_blocks"], "ident", d["check"]["handles_bit_identical"][:5])
PY
}
# 1. the driver's exact command, first thing on the fresh box (cold image, cold clocks)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/${tag}_driver.json 2> gpurun_out/r6/${tag}_driver.err
sum "${tag} driver-cmd" gpurun_out/r6/${tag}_driver.json
for rep in 1 2; do
  for d in $depths; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-profile --cpu-instances 0 --no-fma-variant --inflight $d > gpurun_out/r6/${tag}_d${d}_${rep}.json 2>/dev/null
    sum "${tag} depth=$d rep=$rep" gpurun_out/r6/${tag}_d${d}_${rep}.json
  done
done
