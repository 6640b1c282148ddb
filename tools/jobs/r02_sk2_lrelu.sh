cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
SK2_SHAPES="stage0 k11,stage0 k3,stage1 k11,stage1 k3,up1,stage2 k11,stage2 k7,stage2 k3" timeout 900 python tools/sk2_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_bench_lrelu.txt
SK2_LRELU=1 SK2_SHAPES="stage0 k11,stage0 k3,stage1 k11,stage1 k3,up1,stage2 k11,stage2 k7,stage2 k3" timeout 900 python tools/sk2_bench.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02/sk2_bench_lrelu.txt
