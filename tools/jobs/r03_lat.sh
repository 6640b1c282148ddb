cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
python tools/latency_breakdown.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03/latency_breakdown.txt
