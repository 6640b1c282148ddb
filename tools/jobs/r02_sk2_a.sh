cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_ops_gpu.py -q -k "stream_k2" 2>&1 | tail -15 > gpurun_out/r02/sk2_tests.log
cat gpurun_out/r02/sk2_tests.log
timeout 900 python tools/sk2_bench.py tools/libss_k2s0.so tools/libss_k2s2.so 2>&1 | tee gpurun_out/r02/sk2_bench_a.txt
