#!/bin/bash
# dev: counters of the attention kernels of scripts/dev/attn_layers.py, one rocprofv3 --pmc pass per counter group.
# usage: scripts/dev/attn_counters.sh <tag> [attn_layers.py args]     -> gpurun_out/<tag>/counters.txt
tag=${1:-attn}; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d $out/pass_$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/dev/attn_layers.py "$@" > $out/pass_$i.log 2>&1 || tail -5 $out/pass_$i.log
done
cd $GRAFT_REPO_ROOT
python - "$out" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/pass_*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        if "edge_attn" not in r["Kernel_Name"]:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"], r["Counter_Name"]) not in seen:
            seen.add((k, r["Dispatch_Id"], r["Counter_Name"])); n[k][r["Counter_Name"]] += 1
with open(out + "/counters.txt", "w") as fh:
    for k, c in sorted(agg.items()):
        line = k + "\n   " + "  ".join(f"{name} {v / max(n[k][name], 1):.4g}" for name, v in sorted(c.items()))
        fh.write(line + "\n"); print(line)
PY
rm -rf $out/pass_*/
