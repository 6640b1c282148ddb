cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/t; mkdir -p $O
timeout 600 python -m pytest tests/test_edge_gpu.py -q -m gpu > $O/edge.log 2>&1; tail -6 $O/edge.log
