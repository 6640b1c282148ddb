cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/b; mkdir -p $O
( timeout 600 python -m pytest tests/test_batch_gpu.py -x -q -k "fused" ) > $O/test_fused.log 2>&1; tail -15 $O/test_fused.log
( timeout 600 python tools/resblock_bench.py 32 ) 2>&1 | grep -v amdgpu.ids | tee $O/resblock_bench.txt
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-streaming-line --no-multilingual --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'P'
import json
d=json.load(open('gpurun_out/r03/b/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_second_kernel'] and (d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['achieved']), d['latency_ms_single_stream'], d['bf16x3']['value'])
P
