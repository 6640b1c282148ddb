cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
SK2_SHAPES="stage0 k11,stage1 k3,unit fc2" timeout 900 python tools/sk2_bench.py tools/libss_k2abl32.so tools/libss_k2abl64.so tools/libss_k2abl128.so tools/libss_k2abl96.so tools/libss_k2abl224.so 2>&1 | grep -v amdgpu.ids | grep -v "^shape" | tee gpurun_out/r02/sk2_ablation_epilogue.txt
