cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "glu or linear_shapes or ln_linear" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_stages_gpu.py -x -q -m gpu -k "incremental_encoder" 2>&1 | tail -2
for sec in 5.26 8.0 2.5; do SS_B1_SECONDS=$sec python tools/b1_profile.py 40 2>/dev/null | tail -1; done
