cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout 300 python -m pytest tests/test_batch_gpu.py tests/test_edge_gpu.py -q 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --streams 1 --no-latency-pass --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms/step', d['ms_per_step'], 'RTFx', d['value'])
for k in ('roofline','roofline_second_kernel'):
    r=d[k]; print(r['kernel'], 'TF', r['achieved'], 'avg us', r['avg_launch_us'], 'n', r['launches'], 'kernel_time/wall', r['kernel_time_over_wall'])"
