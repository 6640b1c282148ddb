cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02/soak; mkdir -p $O
timeout 600 python bench.py --steps 400 --warmup 5 --no-latency-pass --no-cpu-baseline > $O/b.json 2> $O/b.err; tail -2 $O/b.err
python -c "import json; d=json.load(open('$O/b.json')); print(d['value'], d['ms_per_step'], d['stream_k_spin_timeouts'], d['roofline']['frac'], d['bf16x3']['value'], d['config']['workload'][:80])"
