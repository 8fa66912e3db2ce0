#!/bin/bash
# Does the attention gather's fabric traffic depend on how many workgroups are resident?  LS_EDGE_LDS_PAD = unused dynamic LDS.
cd /tmp && export TMPDIR=/tmp
for pad in 0 30000 100000; do
  out=$GRAFT_REPO_ROOT/gpurun_out/attn_occ_$pad; mkdir -p $out
  LS_EDGE_LDS_PAD=$pad rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/pmc_ops.py --manifest $out/manifest.json > $out/log.txt 2>&1
  python - $out $pad <<'PY'
import csv, glob, sys
out, pad = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0]
tot = {}
for r in csv.DictReader(open(f)):
    if "edge_attn_v4" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        k = r["Kernel_Name"][:40] + " grid " + r["Grid_Size"]
        tot.setdefault(k, []).append(float(r["Counter_Value"]))
for k, v in tot.items(): print(f"pad {pad}: {k}: FETCH {sum(v)/len(v)*64*2/1e6:.0f} MB per launch ({len(v)} launches)")
PY
  LS_EDGE_LDS_PAD=$pad python $GRAFT_REPO_ROOT/scripts/pmc_ops.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pad $pad us:', {k:v for k,v in d.items() if 'edge_attn' in k})"
done
