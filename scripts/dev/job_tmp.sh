cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t7.log
python bench.py > gpurun_out/b7.json 2> gpurun_out/b7.err
python bench.py --steps 20 --warmup 5 > gpurun_out/b7_20.json 2> /dev/null
python __graft_entry__.py --smoke > gpurun_out/smoke7.log 2>&1
