# Round-4 (e): conv_c64 (64-channel vocoder stage: slab + streamed weights) -- unit tests, micro-benchmark, whole suite, bench A/B.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/e; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c64" ) > $O/pytest_c64.log 2>&1; tail -5 $O/pytest_c64.log
timeout 600 python tools/c64_bench.py > $O/c64_bench.txt 2>&1; cat $O/c64_bench.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['frac'], 'lat', d['latency_ms_single_stream'], d['process_census'].get('conv_c64<256,64>'))"; tail -2 $O/bench.err
SS_NO_CONV_C64=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe > $O/bench_noc64.json 2> $O/bench_noc64.err; python -c "import json; d=json.load(open('$O/bench_noc64.json')); print('bench (stage on conv_sk2<64>):', d['value'], d['ms_per_step'], d['roofline']['frac'])"
