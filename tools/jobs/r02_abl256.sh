cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
export SK2_SHAPES="stage0 k11,stage0 k3,stage1 k7,stage2 k11,stage2 k3"
timeout 900 python tools/sk2_bench.py tools/libss_k2abl256.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_ablation_apieces.txt
