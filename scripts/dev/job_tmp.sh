cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python scripts/dev/build_variants.py w2m16:gemm.hip=-DLS_W2_MFMA16 > /dev/null
bash scripts/dev/w2_clock.sh w2m16 > gpurun_out/w2_clock_r5.txt 2>&1
LS_TAG=product python scripts/dev/w2_time.py >> gpurun_out/w2_time_r5.txt 2>&1
LS_TAG=mfma16x16x32 LS_LIB_PATH=$PWD/livingscenes_amd/lib/variants/w2m16/liblivingscenes_hip.so python scripts/dev/w2_time.py >> gpurun_out/w2_time_r5.txt 2>&1
LS_TAG=product python scripts/dev/w2_time.py >> gpurun_out/w2_time_r5.txt 2>&1
bash scripts/dev/marginal_cost.sh marg_r5 > /dev/null 2>&1
