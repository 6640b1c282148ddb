# Round-4 (j): conv_c32 A/B on ONE stream (kernel time without cross-stream effects) and with pair fusion kept for k = 3
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/j; mkdir -p $O
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --streams 1"
for v in 1 2; do
timeout 900 python bench.py $X > $O/bench_1s.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_1s.json')); print('1 stream, conv_c32:', d['value'], d['ms_per_step'])"
SS_NO_CONV_C32=1 timeout 900 python bench.py $X > $O/bench_1s_noc32.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_1s_noc32.json')); print('1 stream, fused ResBlocks:', d['value'], d['ms_per_step'])"
done
python tools/resblock_bench.py 32 > $O/resblock_bench.txt 2>&1; tail -12 $O/resblock_bench.txt
