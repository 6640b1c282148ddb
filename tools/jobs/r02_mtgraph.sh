cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/mt_step_graph.py 2>&1 | grep -v amdgpu.ids | tail -30
