#!/bin/bash
# interleaved A/B of the GEMM range scaling (LS_GEMM_RANGE=0: the un-scaled round-2 split): bench, SDF decode, GEMM shapes
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2 3; do
  for mode in 1 0; do
    LS_GEMM_RANGE=$mode python bench.py --cpu-instances 0 --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('range=$mode', 'bench', round(d['value']), d['ms_per_step'])" >> $out/ab.log
    LS_GEMM_RANGE=$mode python scripts/sdf_microbench.py 2>/dev/null | grep "sdf decode" | sed "s/^/range=$mode /" >> $out/ab.log
  done
done
python scripts/gemm_microbench.py 2>/dev/null | grep -v amdgpu.ids > $out/gemm.log
LS_GEMM_RANGE=0 python scripts/gemm_microbench.py 2>/dev/null | grep -v amdgpu.ids > $out/gemm_noscale.log
cat $out/ab.log; cat $out/gemm.log; cat $out/gemm_noscale.log
