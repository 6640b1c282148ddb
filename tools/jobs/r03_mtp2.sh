cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/mtp2; mkdir -p $O
( time timeout 900 python -m pytest tests/test_mt_persistent_gpu.py -x -q ) > $O/tests.log 2>&1; tail -6 $O/tests.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03/mtp2/bench.json'))
print(d['value'], d['latency_ms_single_stream'], d['latency_ms_single_stream_persistent_mt_step'], d['stream_k_spin_timeouts'])
s=d['streaming_320ms']
for k in ('incremental','full_recompute','incremental_launch_per_op_mt'):
    print(k, s[k]['rtfx_compute'], s[k]['ms_per_policy_call_mean'], s[k]['ms_per_policy_call_p95'], s[k]['gemm_class_launches_per_policy_call'])
P
