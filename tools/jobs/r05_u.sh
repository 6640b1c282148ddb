# Round-5 (u): 200-step soak; the typed two-rank command on the one GPU over gloo (weak and strong) with the round-5 line (placement fields)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/u; mkdir -p $O
X="--no-latency-pass --no-cpu-baseline --no-bf16x3-line --no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe --no-soak"
timeout 600 python bench.py --steps 200 --warmup 10 $X > $O/soak_200_steps.json 2> $O/soak.err; python -c "import json; d=json.load(open('$O/soak_200_steps.json')); print('soak 200 steps:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d.get('stream_k_spin_timeouts'))"
SS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 $X > $O/two_ranks_one_gpu_gloo.json 2> $O/two.err; wc -l $O/two_ranks_one_gpu_gloo.json; python -c "import json; d=json.load(open('$O/two_ranks_one_gpu_gloo.json')); print(d['value'], d['n_gpus'], d['per_rank'], d['self_launched'])"
SS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --scaling strong $X > $O/two_ranks_one_gpu_strong_gloo.json 2> $O/two_s.err; python -c "import json; d=json.load(open('$O/two_ranks_one_gpu_strong_gloo.json')); print(d['value'], d['n_gpus'], d['per_rank'])"
SS_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 5 $X > $O/bench_force_dist.json 2> $O/fd.err; python -c "import json; d=json.load(open('$O/bench_force_dist.json')); print(d['value'], d['rccl'])"
