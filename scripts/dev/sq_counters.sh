#!/bin/bash
# dev: SQ wait / activity counters of every kernel of scripts/pmc_ops.py whose name matches a pattern (one rocprofv3 --pmc pass).
# usage: scripts/dev/sq_counters.sh <tag> <pattern>     -> gpurun_out/<tag>/sq.txt
tag=${1:-sq}; pat=${2:-edge_ft}
out=$PWD/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d $out/pass -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/pmc_ops.py --reps 2 > $out/log.txt 2>&1 || tail -5 $out/log.txt
cd $GRAFT_REPO_ROOT
python - "$out" "$pat" <<'PY'
import csv, glob, re, sys, collections
out, pat = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(out + "/pass/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in rows:
    if re.search(pat, r["Kernel_Name"]):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
with open(out + "/sq.txt", "w") as fh:
    for k, c in agg.items():
        wc = c["SQ_WAVE_CYCLES"] or 1
        line = (f"{k}\n   launches {n[k]}  wave-cycles(quad) {wc/n[k]:.3g}  wait_any {c['SQ_WAIT_ANY']/wc:.2f}  wait_inst_any {c['SQ_WAIT_INST_ANY']/wc:.2f}  "
                f"active_any {c['SQ_ACTIVE_INST_ANY']/wc:.2f}  active_valu {c['SQ_ACTIVE_INST_VALU']/wc:.2f}  active_lds {c['SQ_ACTIVE_INST_LDS']/wc:.2f}  "
                f"valu_insts/launch {c['SQ_INSTS_VALU']/n[k]:.3g}  mfma_busy_cycles/launch {c['SQ_VALU_MFMA_BUSY_CYCLES']/n[k]:.3g}  "
                f"mfma_busy_frac {c['SQ_VALU_MFMA_BUSY_CYCLES']/max(c['GRBM_GUI_ACTIVE']/8*1024,1):.3f}  gui_active/launch {c['GRBM_GUI_ACTIVE']/8/n[k]:.3g}")
        fh.write(line + "\n"); print(line)
PY
rm -rf $out/pass
