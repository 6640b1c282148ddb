cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
timeout 300 python -m pytest tests/test_reference_agent_gpu.py tests/test_stages_gpu.py -q -k "agent" 2>&1 | tail -3
for seg in 320 640; do
timeout 600 python bench.py --mode streaming --utterances 12 --segment-ms $seg > gpurun_out/r02/stream_bench_$seg.json 2> gpurun_out/r02/stream_bench_$seg.err || tail -3 gpurun_out/r02/stream_bench_$seg.err
done
python - <<'PY'
import json
for f in ("gpurun_out/r02/stream_bench_320.json", "gpurun_out/r02/stream_bench_640.json"):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "failed", e); continue
    print(f, d["value"])
    for k in ("incremental", "full_recompute"):
        print(" ", k, d[k])
PY
