cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -m pytest tests/test_hip_layers.py tests/test_hip_fullbatch.py tests/test_hip_parity.py tests/test_hip_range.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t11.log
bash scripts/dev/prof_kernels.sh prof1fl . --inflight 1 > /dev/null 2>&1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],4), d['check']['handles_bit_identical'][:5])"; }
python bench.py --cpu-instances 0 --no-fma-variant --no-profile --inflight 1 2>/dev/null | tail -1 | line rel_1fl >> gpurun_out/ab11.log
python bench.py --cpu-instances 0 --no-fma-variant --no-profile --steps 20 --warmup 5 2>/dev/null | tail -1 | line rel_20 >> gpurun_out/ab11.log
python bench.py --cpu-instances 0 --no-fma-variant --no-profile 2>/dev/null | tail -1 | line rel >> gpurun_out/ab11.log
