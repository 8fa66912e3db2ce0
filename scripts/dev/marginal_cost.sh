#!/bin/bash
# marginal cost of each kernel family in the 12-in-flight bench: skip the family's launches after the warm-up (stale results; timing only)
# the skip knob exists only in a variant library built with -DLS_DEV_KNOBS (the release library ignores LS_SKIP)
out=gpurun_out/$1; mkdir -p $out
python scripts/dev/build_variants.py devknobs:model.hip=-DLS_DEV_KNOBS > /dev/null || exit 1
export LS_LIB_PATH=$PWD/livingscenes_amd/lib/variants/devknobs/liblivingscenes_hip.so
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), d['ms_per_step'])"; }
python bench.py --cpu-instances 0 --no-profile --no-fma-variant 2>/dev/null | tail -1 | line none >> $out/marginal.log
for fam in knn attn pool l0 tables glob fps tail prologue "knn,attn" "knn,attn,tables,glob" "knn,attn,tables,glob,fps,pool,l0,tail,prologue"; do
  LS_SKIP=$fam python bench.py --cpu-instances 0 --no-profile --no-fma-variant 2>/dev/null | tail -1 | line $fam >> $out/marginal.log
done
python bench.py --cpu-instances 0 --no-profile --no-fma-variant 2>/dev/null | tail -1 | line none >> $out/marginal.log
for fam in knn attn tables glob fps; do
  LS_SKIP=$fam python bench.py --cpu-instances 0 --no-profile --no-fma-variant --inflight 1 2>/dev/null | tail -1 | line "1fl:$fam" >> $out/marginal.log
done
python bench.py --cpu-instances 0 --no-profile --no-fma-variant --inflight 1 2>/dev/null | tail -1 | line "1fl:none" >> $out/marginal.log
cat $out/marginal.log
