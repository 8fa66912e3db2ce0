#!/bin/bash
# _solve_end2end(mesh=False) per scene pair (encode_fps x2, matcher, Kabsch, ICP): wall time vs summed kernel time
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/e2e_prof; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/configs_synth.py --scenes 6 --skip-dense --optim-pairs 0 > $out/log.txt 2>&1
grep -v amdgpu $out/log.txt | head -3
python - $out <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"summed kernel time of the whole script: {tot/1e6:.1f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]: print(f'   {r["Name"][:70]:70s} calls {r["Calls"]:>6s} total {float(r["TotalDurationNs"])/1e6:8.1f} ms avg {float(r["AverageNs"])/1e3:8.1f} us')
PY
