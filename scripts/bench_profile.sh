#!/bin/bash
# On the GPU box: the default bench line + the rocprofv3 --kernel-trace --stats summary of the SAME command.
# usage: bash scripts/bench_profile.sh <tag>   -> gpurun_out/<tag>/{bench_line.json, kernel_stats.csv, kernel_stats_top.txt}
set -e
tag=${1:-final}
out=$PWD/gpurun_out/$tag
mkdir -p $out
python bench.py 2>$out/bench.err | tail -1 > $out/bench_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py > $out/bench_under_rocprof.log 2>&1 || tail -5 $out/bench_under_rocprof.log
cd $GRAFT_REPO_ROOT
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
grep "^{\"metric\"" $out/bench_under_rocprof.log | tail -1 > $out/bench_line_under_rocprof.json || true
python - "$out" <<'PY'
import csv, sys
out = sys.argv[1]
rows = sorted(csv.DictReader(open(out + "/kernel_stats.csv")), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out + "/kernel_stats_top.txt", "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats -- python bench.py   (total kernel time {tot/1e6:.1f} ms)\n")
    for r in rows[:40]:
        f.write(f'{r["Name"][:100]:100s} calls {r["Calls"]:>7s}  total {float(r["TotalDurationNs"])/1e6:9.2f} ms  avg {float(r["AverageNs"])/1e3:9.2f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%\n')
print(open(out + "/kernel_stats_top.txt").read())
PY
rm -rf $out/prof
