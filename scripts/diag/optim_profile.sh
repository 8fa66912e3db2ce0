#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/optim_prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/optim_registration_microbench.py 64 2>&1 | grep "P = "
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/optim_prof/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:9.1f} ms  avg {float(r["AverageNs"])/1e3:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY
