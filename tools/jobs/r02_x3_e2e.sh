cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02/x3; mkdir -p $O
timeout 600 python -m pytest tests/test_bf16x3_gpu.py -q 2>&1 | tail -5
timeout 300 python bench.py --steps 20 --warmup 5 --no-latency-pass --no-cpu-baseline > $O/b.json 2> $O/b.err; tail -3 $O/b.err
python -c "import json; d=json.load(open('$O/b.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac']); print(d['bf16x3'])"
