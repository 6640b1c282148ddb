# Round-4 (o): conv_c16 (k = 11 ResBlocks of the 16-channel stage conv by conv, weight matrix in registers): tests, micro-bench, bench A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/o; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c64" ) > $O/pytest_c16.log 2>&1; tail -3 $O/pytest_c16.log
C64_BENCH_CHANNELS=16 timeout 600 python tools/c64_bench.py > $O/c16_bench.txt 2>&1; tail -9 $O/c16_bench.txt
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for v in 1 2; do
timeout 900 python bench.py $X > $O/bench_$v.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_$v.json')); print('8 streams, k=11 of C=16 on conv_c16:', d['value'], d['ms_per_step'])"
SS_NO_CONV_C16=1 timeout 900 python bench.py $X > $O/bench_noc16_$v.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_noc16_$v.json')); print('8 streams, fused ResBlocks at C=16:', d['value'], d['ms_per_step'])"
done
