cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/a; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q -s -k "multilingual or offline_generator or row_max" ) > $O/new_tests.log 2>&1; tail -8 $O/new_tests.log
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -4 $O/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03/a/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['bf16x3']['value'])
print('ml', d['multilingual'])
print('ab', d['event_bracket_perturbation'])
s=d['streaming_320ms']
print('stream', s['value'], s['incremental_speedup_over_full_recompute'], s.get('cpu_baseline'), s.get('long_prefix_sweep'))
P
SS_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 8 --warmup 2 --scaling strong --no-streaming-line > $O/two_ranks_strong.json 2> $O/two_ranks_strong.err; tail -2 $O/two_ranks_strong.err
python -c "import json; d=json.load(open('$O/two_ranks_strong.json')); print(d['value'], d['n_gpus'], d['scaling'], d['steps'], d['steps_per_gpu'], d['per_rank'])"
