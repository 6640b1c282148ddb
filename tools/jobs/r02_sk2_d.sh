cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "stream_k" 2>&1 | tail -5 > gpurun_out/r02/sk2_tests.log
cat gpurun_out/r02/sk2_tests.log
timeout 900 python tools/sk2_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_bench_d.txt
