cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "stream_k" 2>&1 | tail -3
SK2_SHAPES="stage0 k11,stage1 k3,unit fc2,stage2 k11,stage2 k7,stage2 k3,up3" timeout 900 python tools/sk2_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_bench_h.txt
