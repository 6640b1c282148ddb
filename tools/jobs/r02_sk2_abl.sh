cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
export SK2_SHAPES="stage0 k11,stage1 k3,unit fc2"
timeout 900 python tools/sk2_bench.py tools/libss_k2s0.so tools/libss_k2abl1.so tools/libss_k2abl2.so tools/libss_k2abl3.so tools/libss_k2abl4.so tools/libss_k2abl7.so tools/libss_k2abl8.so tools/libss_k2abl15.so tools/libss_k2abl16.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_ablation.txt
