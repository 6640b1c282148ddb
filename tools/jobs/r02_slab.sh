cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
for i in 1 2; do timeout 600 python tools/slab_bench.py "$@" 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r02/slab_bench.txt
