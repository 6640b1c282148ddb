# Round-4 (f): conv_c64, two workgroups per CU, with and without the staggered start of the second half of the grid
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/f; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c64" ) > $O/pytest_c64.log 2>&1; tail -3 $O/pytest_c64.log
timeout 600 python tools/c64_bench.py > $O/c64_bench_stagger.txt 2>&1; cat $O/c64_bench_stagger.txt
SS_CONV_C64_STAGGER=0 timeout 600 python tools/c64_bench.py > $O/c64_bench_nostagger.txt 2>&1; cat $O/c64_bench_nostagger.txt
SS_CONV_C64_STAGGER=340 timeout 600 python tools/c64_bench.py > $O/c64_bench_stagger340.txt 2>&1; cat $O/c64_bench_stagger340.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['frac'], 'lat', d['latency_ms_single_stream'])"; tail -2 $O/bench.err
