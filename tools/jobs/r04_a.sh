# Round-4 first check: GPU suite on the housekeeping build, smoke, RCCL probe (world of one), SS_FORCE_DIST=1 bench,
# `bench.py --gpus 2` typed without a launcher (two ranks on the one GPU over gloo), the driver's N = 1 command.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/a; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --rccl-probe > $O/rccl_probe.json 2> $O/rccl_probe.err; cat $O/rccl_probe.json; tail -2 $O/rccl_probe.err
SS_FORCE_DIST=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-bf16x3-line --no-multilingual --no-streaming-line --no-bracket-ab > $O/bench_force_dist.json 2> $O/bench_force_dist.err; python -c "import json; d=json.load(open('$O/bench_force_dist.json')); print('force_dist:', d['value'], d['rccl'])"; tail -2 $O/bench_force_dist.err
SS_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/two_ranks_self_launched.json 2> $O/two_ranks.err; echo "rc=$?"; python -c "import json; d=json.load(open('$O/two_ranks_self_launched.json')); print('two ranks self-launched:', d['value'], d['n_gpus'], d['self_launched'], d['per_rank'], d['rccl'])"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('driver cmd:', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_second_kernel']['kernel'], d['rccl'])"
