# Round-4 (m): MT search as one persistent launch (device-side token loop): tests + latency A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/m; mkdir -p $O
( timeout 900 python -m pytest tests/test_mt_persistent_gpu.py tests/test_stages_gpu.py tests/test_edge_gpu.py tests/test_reference_agent_gpu.py tests/test_offline_generator_gpu.py -q -x ) > $O/pytest_mt.log 2>&1; tail -5 $O/pytest_mt.log
timeout 600 python tools/latency_breakdown.py > $O/latency.txt 2>&1; tail -12 $O/latency.txt
SS_NO_MT_DEVICE_LOOP=1 timeout 600 python tools/latency_breakdown.py > $O/latency_per_token.txt 2>&1; tail -12 $O/latency_per_token.txt
