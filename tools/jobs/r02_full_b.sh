# Round-2 evidence run: GPU tests, smoke, the driver's bench command, rocprofv3 kernel stats of the same command and of
# the one-stream form, three separate PMC passes (FETCH_SIZE / WRITE_SIZE / MfmaUtil) as MI355X_MICROARCH.md prescribes.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02/full_b
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest_gpu.log 2>&1
tail -6 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python bench.py --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/prof_driver.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1stream -- python bench.py --steps 20 --warmup 5 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line > $O/bench_1stream_under_rocprof.json 2> $O/prof_1stream.err
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line > $O/pmc_$c.bench.json 2> $O/pmc_$c.err
done
F=$(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1); U=$(ls $O/pmc_MfmaUtil/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $F $W $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc MfmaUtil (three separate passes, --kernel-trace only) of bench.py --steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line; gfx950 correction: HBM read bytes = 2*FETCH_SIZE KB, WRITE_SIZE as reported; algorithmic bytes = the library's census of the same process" $U $O/pmc_FETCH_SIZE.bench.json > $O/pmc_classes.txt 2>&1
cat $O/pmc_classes.txt | head -40
rm -f $O/pmc_*/*/*counter_collection.csv $O/*/*/*kernel_trace.csv     # large; the summaries stay
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print({k: d[k] for k in ("value","utterances_per_sec","ms_per_step","latency_ms_single_stream")})
for k in ("roofline","roofline_second_kernel"):
    r = d[k]; print(r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r["launches"], r.get("in_region"))
print(d["cpu_baseline"])
PY
