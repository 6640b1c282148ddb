# Round-4 evidence run: the driver's bench command, rocprofv3 kernel stats of the same command and of the one-stream form,
# three separate PMC passes (FETCH_SIZE / WRITE_SIZE / MfmaUtil) as MI355X_MICROARCH.md prescribes.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/full
mkdir -p $O
export TMPDIR=/tmp
X="--no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
# rocprofv3 7.2 often dies (SIGSEGV inside its own HIP-API interception) under the 8 host threads of this command: up to 3 tries
for try in 1 2 3; do
  rm -rf $O/prof_driver
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python bench.py --steps 20 --warmup 5 > $O/bench_under_rocprof.json 2> $O/prof_driver.err && ls $O/prof_driver/*/*_kernel_stats.csv > /dev/null 2>&1 && break
  echo "rocprofv3 on the driver command: try $try failed"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1stream -- python bench.py --steps 20 --warmup 5 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X > $O/bench_1stream_under_rocprof.json 2> $O/prof_1stream.err
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X > $O/pmc_$c.bench.json 2> $O/pmc_$c.err
done
F=$(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1); U=$(ls $O/pmc_MfmaUtil/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $F $W $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc MfmaUtil (three separate passes, --kernel-trace only) of bench.py --steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X; gfx950 correction: HBM read bytes = 2*FETCH_SIZE KB, WRITE_SIZE as reported; algorithmic bytes = the library's census of the same process" $U $O/pmc_FETCH_SIZE.bench.json > $O/pmc_classes.txt 2>&1
cat $O/pmc_classes.txt | head -60
rm -f $O/pmc_*/*/*counter_collection.csv $O/*/*/*kernel_trace.csv     # large; the summaries stay
python tools/share_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/bench_1stream_under_rocprof.json > $O/share_table.md; cat $O/share_table.md
python tools/roofline_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/pmc_traffic.json > $O/roofline_table.md; cat $O/roofline_table.md
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print({k: d[k] for k in ("value","utterances_per_sec","ms_per_step","latency_ms_single_stream")})
for k in ("roofline","roofline_second_kernel"):
    r = d[k]; print(r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r["launches"], r.get("in_region"))
print(d["cpu_baseline"]["value"], d["multilingual"]["value"], d["streaming_320ms"]["value"])
PY
