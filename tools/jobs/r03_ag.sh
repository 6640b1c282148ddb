cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
timeout 120 ./tools/bin/allgather_probe 2>&1 | tee gpurun_out/r03/allgather_probe.txt
