# Round-5 (s): the whole GPU suite + smoke on the committed build (after the duration-predictor change)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/s; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
