cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/c; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03/c/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_detail'] and d['roofline']['traffic_detail'].get('same_kernel_sources'), d['cpu_baseline']['value'], d['bf16x3']['value'], d['latency_ms_single_stream'])
print('ml', d['multilingual']['value'], 'ab', d['event_bracket_perturbation']['with_over_without'], 'stream', d['streaming_320ms']['value'], d['stream_k_spin_timeouts'])
P
