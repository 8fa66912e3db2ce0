cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python scripts/dev/build_variants.py ftv3:edge_fused.hip=-DLS_FT_V_WPE=3 > /dev/null
bash scripts/dev/ab_libs.sh ftv3 > gpurun_out/ab_ftv3.log 2>&1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],4), d['check']['handles_bit_identical'][:5])"; }
for v in rel ftv3 rel ftv3; do
  [ $v = rel ] && unset LS_LIB_PATH || export LS_LIB_PATH=$PWD/livingscenes_amd/lib/variants/$v/liblivingscenes_hip.so
  python bench.py --cpu-instances 0 --no-fma-variant --no-profile --inflight 1 2>/dev/null | tail -1 | line ${v}_1fl >> gpurun_out/ab_ftv3.log
done
