cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python scripts/configs_synth.py > gpurun_out/configs_synth_r5.log 2>&1
