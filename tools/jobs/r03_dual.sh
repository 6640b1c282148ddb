cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
./tools/bin/dual_pipe_rate 2>&1 | tee gpurun_out/r03/dual_pipe_rate.txt
