# Key-split rel-pos attention: op tests, encoder tests that go through it, B = 1 latency and the streaming line with / without.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/attn; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_stages_gpu.py tests/test_edge_gpu.py tests/test_reference_agent_gpu.py -x -q -m gpu -k "encoder or minimal or agent" 2>&1 | tail -3
X="--no-cpu-baseline --no-bf16x3-line --no-multilingual --no-bracket-ab"
for mode in off on; do
  if [ $mode = off ]; then export SS_ATTN_NO_SPLIT=1; else unset SS_ATTN_NO_SPLIT; fi
  timeout 600 python bench.py --steps 6 --warmup 2 $X > $O/bench_$mode.json 2> $O/bench_$mode.err
  python - <<PY
import json
d = json.load(open("$O/bench_$mode.json"))
s = d["streaming_320ms"]
print("$mode", d["value"], d["latency_ms_single_stream"], d["latency_ms_single_stream_persistent_mt_step"], s["value"], s["incremental"]["ms_per_policy_call_mean"], s["incremental"]["ms_per_policy_call_p95"])
for r in s["long_prefix_sweep"]:
    print("   ", r["source_s"], r["incremental"]["encoder_side_ms_total"], r["full_recompute"]["encoder_side_ms_total"], r["speedup_encoder_side"], r.get("speedup_total"))
PY
done
