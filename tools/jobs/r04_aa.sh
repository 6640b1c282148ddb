# Round-4 (aa): the N > 1 path of the committed build on the one GPU there is: `bench.py --gpus 2` typed without a launcher (two ranks, gloo,
# same device), weak and strong scaling; the N = 1 line through a live RCCL communicator (SS_FORCE_DIST=1)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/aa; mkdir -p $O
X="--no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-bracket-ab --no-latency-pass"
SS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 $X > $O/two_ranks_weak.json 2> $O/two_ranks_weak.err; echo "rc=$?"
python -c "import json; d=json.load(open('$O/two_ranks_weak.json')); print('weak:', d['value'], d['n_gpus'], d['self_launched'], d['scaling'], [(p['rank'], p['utterances']) for p in d['per_rank']], d['rccl'])"
SS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 --scaling strong $X > $O/two_ranks_strong.json 2> $O/two_ranks_strong.err; echo "rc=$?"
python -c "import json; d=json.load(open('$O/two_ranks_strong.json')); print('strong:', d['value'], d['n_gpus'], d['scaling'], [(p['rank'], p['utterances']) for p in d['per_rank']])"
SS_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $X > $O/force_dist.json 2> $O/force_dist.err; echo "rc=$?"
python -c "import json; d=json.load(open('$O/force_dist.json')); print('force dist:', d['value'], d['rccl'])"
