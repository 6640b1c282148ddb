# Round-5 evidence run at the measured configuration (packs of 128): GPU suite + smoke, the measured-configuration parity log, the
# driver's bench command, rocprofv3 kernel stats of the same command and of the one-stream form, three separate PMC passes
# (FETCH_SIZE / WRITE_SIZE / MfmaUtil) as MI355X_MICROARCH.md prescribes, then the bench once more so that its line joins the
# PMC summary of THIS build.  (Instruction-mix counters and the phase accounting of the Winograd kernels do not depend on the pack:
# profiles/r05_pmc_extra.md, r05_cw_timing.txt stay.)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/final
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
X="--no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe --no-soak"
for try in 1 2 3; do
  rm -rf $O/prof_driver
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bf16x3-line --no-latency-pass $X > $O/bench_under_rocprof.json 2> $O/prof_driver.err && ls $O/prof_driver/*/*_kernel_stats.csv > /dev/null 2>&1 && break
  echo "rocprofv3 on the 8-stream command: try $try failed"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1stream -- python bench.py --steps 20 --warmup 5 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X > $O/bench_1stream_under_rocprof.json 2> $O/prof_1stream.err
P="--steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X"
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py $P > $O/pmc_$c.bench.json 2> $O/pmc_$c.err
done
F=$(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1); U=$(ls $O/pmc_MfmaUtil/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $F $W $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc MfmaUtil (three separate passes, --kernel-trace only) of bench.py $P; gfx950 correction: HBM read bytes = 2*FETCH_SIZE KB, WRITE_SIZE as reported; algorithmic bytes = the library's census of the same process" $U $O/pmc_FETCH_SIZE.bench.json > $O/pmc_classes.txt 2>&1
head -30 $O/pmc_classes.txt
rm -f $O/pmc_*/*/*counter_collection.csv $O/*/*/*kernel_trace.csv
python tools/share_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/bench_1stream_under_rocprof.json > $O/share_table.md; cat $O/share_table.md
python tools/roofline_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/pmc_traffic.json > $O/roofline_table.md; head -40 $O/roofline_table.md
cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json      # the bench line below joins the PMC summary of this build
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print({k: d[k] for k in ("value","utterances_per_sec","ms_per_step","latency_ms_single_stream","near_tie_rows","hbm")})
for k in ("roofline","roofline_second_kernel"):
    r = d[k]; print(r["kernel"], r["achieved"], r["frac"], r.get("frac_issued"), r["avg_launch_us"], r["launches"], r["traffic"], r["traffic_over_algorithmic"], r["traffic_detail"]["same_kernel_sources"] if r.get("traffic_detail") else None)
print(d["roofline_family"]["frac"], d["roofline_family"]["frac_issued"], d["soak"]["value"], d["pack_invariance"]["alone_equals_in_pack_bitwise"])
print(d["cpu_baseline"]["value"], d["multilingual"]["value"], d["streaming_320ms"]["value"], d["bf16x3"]["value"])
PY
