#!/bin/bash
# A/B of the timing variants (scripts/dev/build_variants.py): bench, GEMM and SDF micro-benchmarks per variant
out=gpurun_out/$1; shift
mkdir -p $out
V=livingscenes_amd/lib/variants
for name in default "$@"; do
  if [ $name = default ]; then unset LS_LIB_PATH; else export LS_LIB_PATH=$PWD/$V/$name/liblivingscenes_hip.so; fi
  python bench.py --cpu-instances 0 --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$name', 'bench', round(d['value']), d['ms_per_step'])" >> $out/ab.log
  python scripts/gemm_microbench.py 2>/dev/null | grep -v amdgpu.ids | sed "s/^/$name /" >> $out/ab_gemm.log
  python scripts/sdf_microbench.py 2>/dev/null | grep "sdf decode" | sed "s/^/$name /" >> $out/ab.log
done
unset LS_LIB_PATH
LS_GEMM_RANGE=0 python bench.py --cpu-instances 0 --no-profile 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('noscale', 'bench', round(d['value']), d['ms_per_step'])" >> $out/ab.log
LS_GEMM_RANGE=0 python scripts/gemm_microbench.py 2>/dev/null | grep -v amdgpu.ids | sed "s/^/noscale /" >> $out/ab_gemm.log
LS_GEMM_RANGE=0 python scripts/sdf_microbench.py 2>/dev/null | grep "sdf decode" | sed "s/^/noscale /" >> $out/ab.log
cat $out/ab.log
