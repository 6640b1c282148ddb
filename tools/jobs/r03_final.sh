# Round-3 final check: full GPU suite, smoke, the driver's bench command, two ranks on the one GPU (gloo), then the evidence
# passes of tools/jobs/r03_full.sh (rocprofv3 kernel stats + three PMC passes) on the same build.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/final; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SS_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 2> $O/two_ranks.err | grep '^{"metric"' > $O/two_ranks.json; python -c "import json; d=json.load(open('$O/two_ranks.json')); print('two ranks:', d['value'], d['n_gpus'], d['scaling'], d['per_rank'])"
# sustained throughput: 200 timed steps (6400 utterances, ~6.5 s of GPU time) -- clocks / power settle well inside this
timeout 600 python bench.py --steps 200 --warmup 10 --no-latency-pass --no-cpu-baseline --no-bf16x3-line --no-multilingual --no-streaming-line --no-bracket-ab > $O/soak_200_steps.json 2> $O/soak.err; python -c "import json; d=json.load(open('$O/soak_200_steps.json')); print('soak 200 steps:', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('stream_k_spin_timeouts'))"
bash tools/jobs/r03_full.sh
