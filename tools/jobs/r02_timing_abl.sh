cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
for m in 0 1 2 4 3 7; do echo "== K2_ABL=$m (1 no DMA, 2 no wait+barrier, 4 no ds_read)"; SS_HIP_LIB=$PWD/tools/libss_k2t$m.so timeout 300 python tools/sk2_timing.py 2>&1 | grep -v amdgpu.ids | grep "stage0 k11\|stage2 k11\|stage2 k3"; done | tee gpurun_out/r02/sk2_timing_ablation.txt
