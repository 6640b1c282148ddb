cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
( ./tools/bin/mfma_lds_rate; ./tools/bin/mfma_lds_rate ) 2>&1 | tee gpurun_out/r03/mfma_lds_rate.txt
