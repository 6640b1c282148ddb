# Round-5 (u2), packs of 128: 200-step soak; the typed two-rank command on the one GPU over gloo (4 streams per rank: two ranks x 8 scratch
# contexts do not fit one GPU); the N = 1 bench through a live RCCL communicator
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/u2; mkdir -p $O
X="--no-latency-pass --no-cpu-baseline --no-bf16x3-line --no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe --no-soak"
timeout 900 python bench.py --steps 200 --warmup 10 $X > $O/soak_200_steps.json 2> $O/soak.err; python -c "import json; d=json.load(open('$O/soak_200_steps.json')); print('soak 200 steps:', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d.get('stream_k_spin_timeouts'), d['config']['utterances_per_gpu'])" || tail -3 $O/soak.err
SS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --streams 4 $X > $O/two_ranks_one_gpu_gloo.json 2> $O/two.err; wc -l $O/two_ranks_one_gpu_gloo.json; python -c "import json; d=json.load(open('$O/two_ranks_one_gpu_gloo.json')); print(d['value'], d['n_gpus'], d['per_rank'], d['self_launched'])" || tail -3 $O/two.err
SS_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 5 $X > $O/bench_force_dist.json 2> $O/fd.err; python -c "import json; d=json.load(open('$O/bench_force_dist.json')); print(d['value'], d['rccl'])" || tail -3 $O/fd.err
