cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/prof1; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1stream -- python bench.py --steps 20 --warmup 5 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line --no-multilingual --no-streaming-line > $O/bench_1stream_under_rocprof.json 2> $O/prof_1stream.err
head -30 $O/prof_1stream/*/*_kernel_stats.csv | cut -c1-170
rm -f $O/*/*/*kernel_trace.csv
python - <<PY
import json
d = json.load(open("$O/bench_1stream_under_rocprof.json"))
print({k: d[k] for k in ("value","utterances_per_sec","ms_per_step")})
PY
