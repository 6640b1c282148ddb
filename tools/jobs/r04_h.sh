# Round-4 (h): conv_c128 -- unit tests, micro-benchmark, bench A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/h; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c64" ) > $O/pytest_c128.log 2>&1; tail -3 $O/pytest_c128.log
C64_BENCH_CHANNELS=128 timeout 600 python tools/c64_bench.py > $O/c128_bench.txt 2>&1; tail -9 $O/c128_bench.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['frac'], d['process_census'].get('conv_c128<192,128>'))"; tail -2 $O/bench.err
SS_NO_CONV_C128=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass > $O/bench_noc128.json 2> $O/bench_noc128.err; python -c "import json; d=json.load(open('$O/bench_noc128.json')); print('bench (stage 128 on conv_sk2):', d['value'], d['ms_per_step'], d['roofline']['frac'])"
