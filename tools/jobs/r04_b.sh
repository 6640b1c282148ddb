# Round-4: fused encoder FFN kernel -- unit tests, micro-benchmark against the two-launch form, then the whole GPU suite and the bench.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/b; mkdir -p $O
( timeout 900 python -m pytest tests/test_ffn_gpu.py -q -x ) > $O/pytest_ffn.log 2>&1; tail -15 $O/pytest_ffn.log
timeout 600 python tools/ffn_bench.py > $O/ffn_bench.txt 2>&1; cat $O/ffn_bench.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-streaming-line --no-multilingual --no-bf16x3-line > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['process_census'].get('ffn_fused<256,2048>'))"
SS_NO_FFN_FUSION=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-streaming-line --no-multilingual --no-bf16x3-line --no-rccl-probe > $O/bench_nofuse.json 2> $O/bench_nofuse.err; python -c "import json; d=json.load(open('$O/bench_nofuse.json')); print('bench (two-launch FFN):', d['value'], d['ms_per_step'], d['roofline']['frac'])"
