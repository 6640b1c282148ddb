# Round-4 (c): margin test + 30-s incremental test + persistent MT step as the primary context's default; FFN micro-benchmark; bench.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/c; mkdir -p $O
( timeout 900 python -m pytest tests/test_margin_gpu.py tests/test_stages_gpu.py -q -x -s -k "margin or 30_second" ) > $O/pytest_new.log 2>&1; grep -E "margins|30-s source|passed|failed|Error|assert" $O/pytest_new.log | head -20
timeout 600 python tools/ffn_bench.py > $O/ffn_bench.txt 2>&1; cat $O/ffn_bench.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['ms_per_step'], d['roofline']['frac'], 'lat', d['latency_ms_single_stream'], d['latency_ms_single_stream_launch_per_op_mt_step'], d['single_stream_mt_step'], 'timeouts', d['stream_k_spin_timeouts']); s=d['streaming_320ms']; print('streaming:', s['value'], s['incremental']['ms_per_policy_call_mean'], s['incremental']['ms_per_policy_call_p95'])"; tail -3 $O/bench.err
