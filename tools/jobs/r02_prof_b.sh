cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_v23 -- python bench.py --steps 20 --warmup 5 --streams 1 --no-latency-pass --no-cpu-baseline > gpurun_out/r02/bench_v23_1stream_rocprof.json 2> gpurun_out/r02/prof_v23.err
head -24 gpurun_out/r02/prof_v23/*/*_kernel_stats.csv | cut -c1-150
python - <<'PY'
import json
d = json.load(open("gpurun_out/r02/bench_v23_1stream_rocprof.json"))
print({k: d[k] for k in ("value","utterances_per_sec","ms_per_step")})
for k in ("roofline", "roofline_second_kernel"):
    r = d[k]; print(r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r["launches"])
PY
