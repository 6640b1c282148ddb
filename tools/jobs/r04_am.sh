# Round-4 (am): committed build (Winograd forms): 200-step soak, the typed two-rank run on the one GPU (gloo), N = 1 through a live RCCL communicator
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/am; mkdir -p $O
X="--no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-bracket-ab --no-latency-pass"
timeout 600 python bench.py --steps 200 --warmup 10 $X --no-rccl-probe > $O/soak_200_steps.json 2> $O/soak.err; python -c "import json; d=json.load(open('$O/soak_200_steps.json')); print('soak 200 steps:', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('stream_k_spin_timeouts'))"
SS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 $X > $O/two_ranks_weak.json 2> $O/two_ranks_weak.err; echo "rc=$? lines=$(wc -l < $O/two_ranks_weak.json)"
python -c "import json; d=json.load(open('$O/two_ranks_weak.json')); print('two ranks:', d['value'], d['n_gpus'], d['self_launched'], [(p['rank'], p['utterances']) for p in d['per_rank']])"
SS_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $X > $O/force_dist.json 2> $O/force_dist.err; echo "rc=$? lines=$(wc -l < $O/force_dist.json)"
python -c "import json; d=json.load(open('$O/force_dist.json')); print('force dist:', d['value'], d['rccl']['backend'], d['rccl']['results_ok'])"
