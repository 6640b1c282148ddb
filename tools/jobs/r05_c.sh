# Round-5 (c): the measured-configuration parity test (strict ids + float64 adjudication) and the margin test (bitwise alone-vs-pack)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/c; mkdir -p $O
timeout 1500 python -m pytest tests/test_bench_config_gpu.py tests/test_margin_gpu.py -q -m gpu -s > $O/tests.log 2>&1; tail -40 $O/tests.log
