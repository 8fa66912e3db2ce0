#!/bin/bash
# round-end artefacts in one lease: bench lines of the three regimes, rocprofv3 kernel stats of the bench command, per-operator PMC passes.
#   bash scripts/dev/final_profiles.sh <tag>     -> gpurun_out/<tag>/...   (copy what is to be judged into profiles/<tag>/)
cd "${GRAFT_REPO_ROOT:-.}"
tag=${1:-r6_final}
out=$PWD/gpurun_out/$tag
mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>$out/driver.err | tail -1 > $out/bench_line_driver_protocol.json
bash scripts/bench_profile.sh $tag > $out/bench_profile.log 2>&1
python bench.py --inflight 1 --steps 100 --cpu-instances 0 --no-fma-variant 2>/dev/null | tail -1 > $out/bench_line_one_in_flight.json
bash scripts/pmc_collect.sh $tag > $out/pmc_collect.log 2>&1
for f in bench_line_driver_protocol bench_line bench_line_one_in_flight; do python - $out/$f.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c = d["config"]; r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "inflight", c["steps_in_flight"], "blocks max/min", c["block_max_over_min"],
      "dominant", r.get("kernel"), "frac", round(r.get("frac", 0), 3), "bound", r.get("bound"))
PY
done
tail -30 $out/pmc_collect.log
