cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
( for w in 1 2; do for z in rand zero; do ./tools/bin/mfma_rate $w $z; done; done; rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 ) 2>&1 | tee gpurun_out/r02/mfma_rate.txt
