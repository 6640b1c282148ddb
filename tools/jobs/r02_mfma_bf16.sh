cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
./tools/bin/mfma_rate_bf16 2>&1 | tee gpurun_out/r02/mfma_rate_bf16.txt
