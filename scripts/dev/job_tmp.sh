cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t_final.log
python __graft_entry__.py --smoke > gpurun_out/smoke_final.log 2>&1
