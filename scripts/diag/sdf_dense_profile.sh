#!/bin/bash
# rocprofv3 kernel stats of the dense SDF decode (configs[4] shape: 8 instances x 128^3 queries per call, 4 calls)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/sdf_prof; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/configs_synth.py --scenes 1 --optim-pairs 0 --dense-instances 32 2>&1 | grep "configs\[4\]"
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/sdf_prof/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total device ms", tot/1e6)
for r in rows[:14]: print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:9.1f} ms  avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
