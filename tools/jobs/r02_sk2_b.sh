cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "stream_k" 2>&1 | tail -5 > gpurun_out/r02/sk2_tests.log
cat gpurun_out/r02/sk2_tests.log
timeout 900 python tools/sk2_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_bench_b.txt
SK2_SHAPES="stage0 k11,stage1 k3,unit fc2,stage0 k3" timeout 900 python tools/sk2_bench.py tools/libss_k2abl8.so tools/libss_k2abl15.so 2>&1 | grep -v amdgpu.ids | grep -v "^library: default" | tee -a gpurun_out/r02/sk2_bench_b.txt
