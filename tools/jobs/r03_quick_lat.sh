# quick look: dwconv / attention op tests, one-utterance latency and the streaming line of the current build
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/quick; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "dwconv or attention" 2>&1 | tail -2
X="--no-cpu-baseline --no-bf16x3-line --no-multilingual --no-bracket-ab"
timeout 600 python bench.py --steps 6 --warmup 2 $X > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
s = d["streaming_320ms"]
print(d["value"], d["latency_ms_single_stream"], d["latency_ms_single_stream_persistent_mt_step"], s["value"], s["incremental"]["ms_per_policy_call_mean"], s["incremental"]["ms_per_policy_call_p95"])
for r in s["long_prefix_sweep"]:
    print("   ", r["source_s"], r["incremental"]["encoder_side_ms_total"], r["full_recompute"]["encoder_side_ms_total"], r["speedup_encoder_side"], r.get("speedup_total"))
PY
