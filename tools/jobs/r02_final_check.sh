cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02/final; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_gpus1.json 2> $O/bench_gpus1.err; tail -2 $O/bench_gpus1.err
python -c "import json; d=json.load(open('$O/bench_gpus1.json')); print(d['value'], d['n_gpus'], d['steps'], d['warmup'], d['scaling'], d['vs_baseline'], d['dtype'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['bf16x3']['value'])"
SS_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 > $O/two_ranks.json 2> $O/two_ranks.err; tail -2 $O/two_ranks.err
python -c "import json; d=json.load(open('$O/two_ranks.json')); print(d['value'], d['n_gpus'], d['per_rank'], d['bf16x3'], d['cpu_baseline'])"
