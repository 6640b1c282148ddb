# Round-4 (ab): stdout carries exactly one line -- N = 1 through a live RCCL communicator (its version banner used to surface after
# the JSON line when stdout is a file), the plain driver command, and the typed two-rank run
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/ab; mkdir -p $O
X="--no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-bracket-ab --no-latency-pass"
SS_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $X > $O/force_dist.json 2> $O/force_dist.err; echo "rc=$? lines=$(wc -l < $O/force_dist.json)"
python -c "import json; d=json.load(open('$O/force_dist.json')); print('force dist:', d['value'], d['rccl'])"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 $X > $O/plain.json 2> $O/plain.err; echo "rc=$? lines=$(wc -l < $O/plain.json)"
python -c "import json; d=json.load(open('$O/plain.json')); print('plain:', d['value'], d['rccl'])"
SS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 $X > $O/two_ranks_weak.json 2> $O/two_ranks_weak.err; echo "rc=$? lines=$(wc -l < $O/two_ranks_weak.json)"
grep -c "RCCL version" $O/force_dist.err $O/plain.err
