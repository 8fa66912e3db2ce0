#!/bin/bash
# dev: rocprofv3 --kernel-trace --stats of a short bench run (one step in flight unless told otherwise), the top kernels printed.
# usage: scripts/dev/prof_kernels.sh <tag> [pattern] [bench args...]      -> gpurun_out/<tag>/kernel_stats_top.txt
tag=${1:-prof}; pat=${2:-.}; shift; shift
out=$PWD/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 420 rocprofv3 --kernel-trace --stats -d $out/prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 8 --no-profile --cpu-instances 0 --no-fma-variant "$@" > $out/under_rocprof.log 2>&1 || tail -5 $out/under_rocprof.log
cd $GRAFT_REPO_ROOT
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
[ -z "$f" ] && { echo "no kernel_stats.csv"; exit 1; }
cp $f $out/kernel_stats.csv
python - "$out" "$pat" <<'PY'
import csv, re, sys
out, pat = sys.argv[1], sys.argv[2]
rows = sorted(csv.DictReader(open(out + "/kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "/kernel_stats_top.txt", "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats (total kernel time {tot/1e6:.1f} ms)\n")
    for r in rows[:60]:
        f.write(f'{r["Name"][:110]:110s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:9.2f} us {100*float(r["TotalDurationNs"])/tot:5.1f}%\n')
for ln in open(out + "/kernel_stats_top.txt"):
    if re.search(pat, ln): print(ln.rstrip())
PY
rm -rf $out/prof
