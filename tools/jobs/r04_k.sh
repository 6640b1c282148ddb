# Round-4 (k): conv_c32 only for the k = 11 ResBlocks of the 32-channel stage: A/B on one stream and on 8
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/k; mkdir -p $O
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for v in 1 2; do
timeout 900 python bench.py $X --streams 1 > $O/b.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/b.json')); print('1 stream, k=11 on conv_c32:', d['value'], d['ms_per_step'])"
SS_NO_CONV_C32=1 timeout 900 python bench.py $X --streams 1 > $O/b.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/b.json')); print('1 stream, fused ResBlocks:', d['value'], d['ms_per_step'])"
timeout 900 python bench.py $X > $O/bench_$v.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_$v.json')); print('8 streams, k=11 on conv_c32:', d['value'], d['ms_per_step'])"
SS_NO_CONV_C32=1 timeout 900 python bench.py $X > $O/bench_noc32_$v.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_noc32_$v.json')); print('8 streams, fused ResBlocks:', d['value'], d['ms_per_step'])"
done
