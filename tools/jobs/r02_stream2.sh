cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
for seg in 320 640 960; do
timeout 600 python bench.py --mode streaming --utterances 12 --segment-ms $seg > gpurun_out/r02/stream_bench_$seg.json 2> gpurun_out/r02/stream_bench_$seg.err || tail -3 gpurun_out/r02/stream_bench_$seg.err
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r02/stream_bench_$seg.json"))
    i = d["incremental"]
    print($seg, d["value"], {k: i[k] for k in ("utterances", "policy_calls", "ms_per_policy_call_mean", "ms_per_policy_call_p95", "writes_per_utterance", "RTF_CA", "utterances_skipped_prefix_over_cap")})
except Exception as e:
    print($seg, "failed", e)
PY
done
