# Round-4 (d): row-tile linear kernel -- unit tests, micro-benchmark, whole GPU suite, bench A/B against the tiled kernel.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/d; mkdir -p $O
( timeout 900 python -m pytest tests/test_rtlin_gpu.py -q -x ) > $O/pytest_rtlin.log 2>&1; tail -6 $O/pytest_rtlin.log
timeout 600 python tools/rtlin_bench.py > $O/rtlin_bench.txt 2>&1; cat $O/rtlin_bench.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['ms_per_step'], d['roofline']['frac'], 'lat', d['latency_ms_single_stream'], d['process_census'].get('rt_linear<48,256>'))"; tail -2 $O/bench.err
SS_NO_RTLIN=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe > $O/bench_nortlin.json 2> $O/bench_nortlin.err; python -c "import json; d=json.load(open('$O/bench_nortlin.json')); print('bench (tiled K=256 linears):', d['value'], d['ms_per_step'], d['roofline']['frac'], 'lat', d['latency_ms_single_stream'])"
