cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout 600 python -m pytest tests/test_batch_gpu.py tests/test_bench_config_gpu.py tests/test_edge_gpu.py -q 2>&1 | tail -3
bash tools/jobs/r02_bench_1s.sh 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('8 streams:', d['value'], d['utterances_per_sec'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'])"
