cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
( time python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -60 ) > gpurun_out/r02/pytest_gpu.log 2>&1
tail -45 gpurun_out/r02/pytest_gpu.log
