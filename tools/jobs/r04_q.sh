# Round-4 (q): conv_c16 restored; which kernel sizes go conv by conv? (k >= 11 default vs k >= 7 at 16 / 32 channels)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/q; mkdir -p $O
( timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c64" ) > $O/pytest_c16.log 2>&1; tail -1 $O/pytest_c16.log
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for v in 1 2; do
timeout 900 python bench.py $X > $O/bench_$v.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_$v.json')); print('default (k >= 11):', d['value'], d['ms_per_step'])"
SS_CONV_C16_MIN_K=7 timeout 900 python bench.py $X > $O/b.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/b.json')); print('C=16: k >= 7:', d['value'], d['ms_per_step'])"
SS_CONV_C32_MIN_K=7 timeout 900 python bench.py $X > $O/b.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/b.json')); print('C=32: k >= 7:', d['value'], d['ms_per_step'])"
done
