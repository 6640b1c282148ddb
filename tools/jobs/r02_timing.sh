cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
SS_HIP_LIB=$PWD/tools/libss_k2timing.so timeout 600 python tools/sk2_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_timing.txt
