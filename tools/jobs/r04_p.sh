# Round-4 (p): k = 11 ResBlocks of the 16-channel stage conv by conv on the existing slab kernel: bench A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/p; mkdir -p $O
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for v in 1 2; do
timeout 900 python bench.py $X > $O/bench_$v.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_$v.json')); print('k=11 of C=16 conv by conv (conv_slab<16>):', d['value'], d['ms_per_step'])"
SS_CONV_C16_MIN_K=99 timeout 900 python bench.py $X > $O/bench_fused16_$v.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_fused16_$v.json')); print('fused ResBlocks at C=16:', d['value'], d['ms_per_step'])"
done
( timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_bench_config_gpu.py tests/test_multilingual_gpu.py -q -x ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
