#!/bin/bash
# On the GPU box: the three counter passes (HBM read, HBM write, SQ activity) of scripts/pmc_ops.py (separate passes, --kernel-trace only: MI355X_MICROARCH.md) + summary.
# usage: bash scripts/pmc_collect.sh <tag>      -> gpurun_out/pmc_<tag>/{pass_*, manifest.json, pmc_latest.json}
set -e
tag=${1:-latest}
root=$PWD/gpurun_out/pmc_$tag
mkdir -p $root
python scripts/pmc_ops.py 2>/dev/null | tail -1 > $root/timings_unprofiled.json     # hipEvent per operator, no profiler attached
cd /tmp && export TMPDIR=/tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $root/pass_$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/pmc_ops.py --manifest $root/manifest.json > $root/pass_$i.log 2>&1 || { tail -5 $root/pass_$i.log; exit 1; }
done
cd $GRAFT_REPO_ROOT
python scripts/pmc_ops_summary.py $root > $root/pmc_latest.json
# keep the merge small: the raw CSVs are large
find $root -name "*.csv" -size +8M -delete
python -c "
import json; d=json.load(open('$root/pmc_latest.json'))
for k,v in d.items():
    if k.startswith('_'): continue
    print(k, round(v.get('hbm_read_bytes',0)/1e6,1), 'MB rd', round(v.get('hbm_write_bytes',0)/1e6,1), 'MB wr', 'mfma', round(v.get('mfma_busy_frac',0),3), 'valu', round(v.get('valu_issue_frac',0),3), d['_hipevent_us'].get(k.replace('layer ','')), 'us')"
