# Round-5 (y): the measured configuration moves to packs of 128 -- parity at that configuration (strict ids, float64 adjudication),
# then the driver's command.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/y; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_bench_config_gpu.py -q -m gpu -s > $O/parity.log 2>&1
grep -E "adjudicated|bench-config parity|passed|failed|Error" $O/parity.log | cut -c1-900 | tail -12
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print({k: d[k] for k in ("value","utterances_per_sec","ms_per_step","latency_ms_single_stream","near_tie_rows","pack_invariance","hbm")})
for k in ("roofline","roofline_second_kernel"):
    r = d[k]; print(r["kernel"], r["achieved"], r["frac"], r.get("frac_issued"), r["avg_launch_us"], r["launches"])
print(d["roofline_family"]["frac"], d["roofline_family"]["frac_issued"], d["soak"]["value"], d["cpu_baseline"]["value"], d["multilingual"])
PY
