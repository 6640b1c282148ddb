cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout 900 python tools/sk2_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/sk2_bench_e.txt
( time timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > gpurun_out/r02/pytest_gpu.log 2>&1
tail -20 gpurun_out/r02/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r02/bench_v22.json 2> gpurun_out/r02/bench_v22.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02/bench_v22.json"))
print({k: d[k] for k in ("value","utterances_per_sec","ms_per_step","latency_ms_single_stream")})
r = d["roofline"]; print(r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r["launches"])
r = d["roofline_second_kernel"]; print(r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r["launches"])
PY
