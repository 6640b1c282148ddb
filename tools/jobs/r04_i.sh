# Round-4 (i): conv_c32 (32-channel vocoder stage as separate slab convs instead of fused ResBlock launches)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/i; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c64" ) > $O/pytest_c32.log 2>&1; tail -3 $O/pytest_c32.log
C64_BENCH_CHANNELS=32 timeout 600 python tools/c64_bench.py > $O/c32_bench.txt 2>&1; tail -9 $O/c32_bench.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['frac'], d['roofline_second_kernel']['avg_launch_us'], d['process_census'].get('conv_c32<256,32>'))"; tail -2 $O/bench.err
SS_NO_CONV_C32=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass > $O/bench_noc32.json 2> $O/bench_noc32.err; python -c "import json; d=json.load(open('$O/bench_noc32.json')); print('bench (fused ResBlocks at C = 32):', d['value'], d['ms_per_step'], d['roofline']['frac'])"
