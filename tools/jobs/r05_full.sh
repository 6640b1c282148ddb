# Round-5 evidence run: GPU suite + smoke, the driver's bench command, rocprofv3 kernel stats of the same command and of the one-stream
# form, three separate PMC passes (FETCH_SIZE / WRITE_SIZE / MfmaUtil) as MI355X_MICROARCH.md prescribes, the instruction-mix counters
# VERDICT r4 #3 asked for (own pass each), phase accounting of the Winograd slab kernels, the measured-configuration parity log.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/full
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python -m pytest tests/test_bench_config_gpu.py tests/test_multilingual_gpu.py tests/test_margin_gpu.py -q -m gpu -s 2>&1 | grep -E "adjudicated|bench-config parity|configs\[4\] parity|pack of|passed|failed" | cut -c1-900 > $O/bench_config_parity.log; tail -3 $O/bench_config_parity.log | cut -c1-300
X="--no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe --no-soak"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
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
head -40 $O/pmc_classes.txt
rocprofv3 -L > $O/avail.txt 2>&1
SPECS=""
for c in SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES VALUBusy LdsBankConflict MemUnitStalled; do
  if grep -qw "$c" $O/avail.txt; then
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py $P > $O/pmc_$c.bench.json 2> $O/pmc_$c.err
    FF=$(ls $O/pmc_$c/*/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$FF" ] && SPECS="$SPECS $c=$FF"
  else
    echo "counter $c not offered by rocprofv3 -L on this box"
  fi
done
python tools/pmc_extra.py $O/pmc_extra.md $SPECS; head -30 $O/pmc_extra.md
rm -f $O/pmc_*/*/*counter_collection.csv $O/*/*/*kernel_trace.csv $O/avail.txt
python tools/share_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/bench_1stream_under_rocprof.json > $O/share_table.md; cat $O/share_table.md
python tools/roofline_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/pmc_traffic.json > $O/roofline_table.md; head -40 $O/roofline_table.md
SS_HIP_LIB=tools/bin/libss_cwt.so timeout 300 python tools/cw_timing.py > $O/cw_timing.txt 2>&1
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print({k: d[k] for k in ("value","utterances_per_sec","ms_per_step","latency_ms_single_stream","near_tie_rows")})
for k in ("roofline","roofline_second_kernel"):
    r = d[k]; print(r["kernel"], r["achieved"], r["frac"], r["avg_launch_us"], r["launches"])
print(d["roofline_family"]["frac"], d["roofline_family"]["frac_issued"], d["soak"]["value"], d["pack_invariance"])
print(d["cpu_baseline"]["value"], d["multilingual"]["value"], d["streaming_320ms"]["value"], d["bf16x3"]["value"])
PY
