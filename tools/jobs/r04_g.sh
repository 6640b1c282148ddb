# Round-4 (g): conv_c64 final form -- whole GPU suite, bench A/B on one box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/g; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
timeout 600 python tools/c64_bench.py > $O/c64_bench.txt 2>&1; tail -8 $O/c64_bench.txt
for v in a b; do
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass > $O/bench_$v.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench_$v.json')); print('bench:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['frac'])"
SS_NO_CONV_C64=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass > $O/bench_noc64_$v.json 2> $O/bench_noc64.err; python -c "import json; d=json.load(open('$O/bench_noc64_$v.json')); print('bench (stage on conv_sk2<64>):', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
