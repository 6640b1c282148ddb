# Round-4 (l): whole GPU suite + smoke + the driver's command on the build with ffn_fused / rt_linear / conv_c64 / conv_c32
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/l; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('driver cmd:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['frac'], 'lat', d['latency_ms_single_stream'], d['latency_ms_single_stream_launch_per_op_mt_step'], 'cpu', d['cpu_baseline']['value'], 'ml', d['multilingual']['value'], 'bf16x3', d['bf16x3']['value'], 'stream', d['streaming_320ms']['value'], d['streaming_320ms']['incremental']['ms_per_policy_call_p95'], d['rccl'])"; tail -2 $O/bench.err
