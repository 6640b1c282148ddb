# Round-5 closing run on the committed tree: GPU suite, the driver's command (its line replaces profiles/r05_bench.json; kernel sources are
# those of the PMC passes of tools/jobs/r05_final.sh), and the two other shapes of the configs[4] leg (packs of 64: a scratch context per
# (language, stream) pair; one stream).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/final2; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
X="--steps 20 --warmup 5 --no-cpu-baseline --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
timeout 600 python bench.py $X --batch 64 > $O/bench_b64.json 2> $O/bench_b64.err; tail -2 $O/bench_b64.err
timeout 600 python bench.py $X --streams 1 --steps 8 --warmup 2 > $O/bench_s1.json 2> $O/bench_s1.err; tail -2 $O/bench_s1.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print({k: d[k] for k in ("value","utterances_per_sec","ms_per_step","latency_ms_single_stream","near_tie_rows","hbm")})
r = d["roofline"]; print(r["kernel"], r["frac"], r.get("frac_issued"), r["traffic"], r["traffic_detail"]["same_kernel_sources"])
print(d["roofline_family"]["frac"], d["soak"]["value"], d["cpu_baseline"]["value"], d["multilingual"]["value"], d["multilingual"]["streams_per_language"], d["streaming_320ms"]["value"], d["bf16x3"]["value"])
for t in ("b64", "s1"):
    e = json.load(open("$O/bench_%s.json" % t)); m = e["multilingual"]
    print(t, e["value"], e["ms_per_step"], m["value"], m.get("contexts"), m.get("streams_per_language"), m.get("multilingual_over_single"))
PY
