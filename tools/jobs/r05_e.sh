# Round-5 (e): the whole GPU suite on the pack-invariant build
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/e; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_bench_config_gpu.py > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
