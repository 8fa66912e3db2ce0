#!/bin/bash
# GPU busy fraction / concurrency of the bench's timed region from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/tr -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 240 --warmup 24 --cpu-instances 0 --no-profile --no-fma-variant > /tmp/bench_tr.json 2>/dev/null
python - <<'PY'
import csv,glob,json
f=glob.glob("/tmp/tr/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# timed region = last 240/264 of the steps: take the last 85% of the time span
t0,t1=rows[0][0],max(r[1] for r in rows)
lo=t1-int(0.40e9)
ev=[]
for s,e,n in rows:
    if e<lo: continue
    ev.append((max(s,lo),1)); ev.append((e,-1))
ev.sort()
busy=0; conc_time={}; cur=0; last=lo
for t,d in ev:
    if cur>0: busy+=t-last
    conc_time[cur]=conc_time.get(cur,0)+(t-last)
    cur+=d; last=t
span=t1-lo
print("span ms",span/1e6,"busy frac",busy/span)
tot=sum(conc_time.values())
print("time share by #kernels in flight:",{k:round(v/tot,3) for k,v in sorted(conc_time.items())[:12]})
d=json.load(open("/tmp/bench_tr.json")); print("bench under tracer:", round(d["value"]), d["ms_per_step"])
PY
