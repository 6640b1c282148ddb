#!/bin/bash
# One parameterised GPU job (replaces the 139 one-letter tools/jobs/rNN_*.sh scripts of rounds 2-5; their history is in git).
#   gpurun --timeout 2400 -- 'bash tools/gpu_job.sh <task> [TAG=r06x] [task arguments]'
# Everything lands under gpurun_out/$TAG/ (scratch); copy what should be judged into profiles/ with tools/collect_evidence.sh.
# Tasks:
#   tests            pytest -m gpu + smoke + the measured-configuration parity log
#   bench [args]     the driver's command (python bench.py [args]) + its sidecar
#   profile          rocprofv3 --kernel-trace --stats of the 8-stream command and of the one-stream form, share / roofline tables
#   pmc              three separate PMC passes (FETCH_SIZE / WRITE_SIZE / MfmaUtil; --kernel-trace only, as MI355X_MICROARCH.md
#                    prescribes) of the one-stream form -> pmc_traffic.json (what bench.py's roofline.traffic reads once copied to profiles/)
#   trace            kernel trace of the timed region (8 streams and 1) -> tools/trace_gaps.py occupancy-in-time tables
#   sweep-streams    bench value / HBM in use at 1 2 4 6 8 streams
#   sweep-packs      bench value at packs of 32 64 128 192 256
#   accuracy [n]     tests/diagnostics/accuracy_vs_float64.py + op_accuracy_gpu.py (they call the CPU oracle, hence under tests/)
#   evidence         tests + bench + profile + pmc + accuracy (the round-end run)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TASK=${1:-evidence}; shift || true
TAG=r06
if [[ "${1:-}" == TAG=* ]]; then TAG=${1#TAG=}; shift; fi
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
QUIET="--no-cpu-baseline --no-multilingual --no-streaming-line --no-rccl-probe --no-soak --no-latency-pass"
bench() { SS_BENCH_DETAIL=$O/$1.detail.json timeout 900 python bench.py "${@:2}" > $O/$1.json 2> $O/$1.err || { echo "bench $1 failed"; tail -3 $O/$1.err; }; }
stats_csv() { ls -t $O/$1/*/*_kernel_stats.csv 2>/dev/null | head -1; }

task_tests() {
  timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
  timeout 900 python -m pytest tests/test_bench_config_gpu.py tests/test_multilingual_gpu.py tests/test_margin_gpu.py -q -m gpu -s 2>&1 \
    | grep -E "adjudicated|bench-config parity|configs\[4\] parity|pack of|passed|failed" | cut -c1-900 > $O/bench_config_parity.log
  tail -3 $O/bench_config_parity.log | cut -c1-300
}
task_bench() {
  bench bench "$@"
  python -c "import json; d=json.load(open('$O/bench.json')); print(len(open('$O/bench.json').read()), 'bytes;', {k: d.get(k) for k in ('value','ms_per_step','soak','multilingual','streaming_320ms','near_tie_rows','hbm_in_use_gb')}); print(d.get('roofline'))"
}
task_profile() {
  for try in 1 2 3; do
    rm -rf $O/prof_driver
    SS_BENCH_DETAIL=$O/bench_under_rocprof.detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python bench.py $QUIET > $O/bench_under_rocprof.json 2> $O/prof_driver.err && [ -n "$(stats_csv prof_driver)" ] && break
    echo "rocprofv3 on the multi-stream command: try $try failed"
  done
  SS_BENCH_DETAIL=$O/bench_1stream_under_rocprof.detail.json rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1stream -- python bench.py --streams 1 $QUIET > $O/bench_1stream_under_rocprof.json 2> $O/prof_1stream.err
  rm -f $O/*/*/*kernel_trace.csv
  python tools/share_table.py "$(stats_csv prof_1stream)" $O/bench_1stream_under_rocprof.detail.json > $O/share_table.md; cat $O/share_table.md
}
task_pmc() {
  P="--steps 6 --warmup 1 --streams 1 $QUIET"
  for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
    SS_BENCH_DETAIL=$O/pmc_$c.detail.json rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py $P > $O/pmc_$c.bench.json 2> $O/pmc_$c.err
  done
  F=$(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1); U=$(ls $O/pmc_MfmaUtil/*/*counter_collection.csv | head -1)
  python tools/pmc_traffic.py $F $W $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc MfmaUtil (three separate passes, --kernel-trace only) of bench.py $P; gfx950 correction: HBM read bytes = 2*FETCH_SIZE KB, WRITE_SIZE as reported; algorithmic bytes = the library's census of the same process" $U $O/pmc_FETCH_SIZE.detail.json > $O/pmc_classes.txt 2>&1
  head -40 $O/pmc_classes.txt
  rm -f $O/pmc_*/*/*counter_collection.csv $O/pmc_*/*/*kernel_trace.csv
  [ -n "$(stats_csv prof_1stream)" ] && python tools/roofline_table.py "$(stats_csv prof_1stream)" $O/pmc_traffic.json > $O/roofline_table.md && head -40 $O/roofline_table.md
}
task_trace() {
  X="$QUIET --no-prof"
  for n in 0 1; do
    S=$([ $n = 1 ] && echo "--streams 1" || echo ""); L=$([ $n = 1 ] && echo 1stream || echo streams)
    SS_BENCH_NO_REPLAY=1 SS_BENCH_DETAIL=$O/trace_$L.detail.json rocprofv3 --kernel-trace --output-format csv -d $O/trace_$L -- python bench.py $X $S > $O/trace_$L.json 2> $O/trace_$L.err
    python tools/trace_gaps.py "$(ls $O/trace_$L/*/*kernel_trace.csv | head -1)" $O/trace_$L.detail.json > $O/trace_gaps_$L.txt 2>&1; cat $O/trace_gaps_$L.txt
  done
  rm -f $O/trace_*/*/*kernel_trace.csv
}
task_sweep_streams() {
  for s in 1 2 4 6 8; do
    bench b_s$s $QUIET --streams $s
    python -c "import json; d=json.load(open('$O/b_s$s.detail.json')); print('streams $s:', d['value'], d['ms_per_step'], d['hbm']['in_use_after_the_timed_region_gb'])"
  done | tee $O/stream_sweep.txt
}
task_sweep_packs() {
  for b in 32 64 128 192 256; do
    bench b_p$b $QUIET --batch $b --steps $((2048 / b))
    python -c "import json; d=json.load(open('$O/b_p$b.detail.json')); print('pack $b:', d['value'], d['ms_per_step'], d['hbm']['in_use_after_the_timed_region_gb'])"
  done | tee $O/pack_sweep.txt
}
task_accuracy() {
  python tests/diagnostics/accuracy_vs_float64.py ${1:-8} > $O/accuracy_vs_float64.json 2> $O/accuracy.err
  python -c "import json; print(json.load(open('$O/accuracy_vs_float64.json'))['summary'])"
  python tests/diagnostics/op_accuracy_gpu.py > $O/op_accuracy.json 2>> $O/accuracy.err
}
case $TASK in
  tests) task_tests ;;
  bench) task_bench "$@" ;;
  profile) task_profile ;;
  pmc) task_pmc ;;
  trace) task_trace ;;
  sweep-streams) task_sweep_streams ;;
  sweep-packs) task_sweep_packs ;;
  accuracy) task_accuracy "$@" ;;
  evidence) task_tests; task_bench; task_profile; task_pmc; task_accuracy 8 ;;
  *) echo "unknown task $TASK"; exit 2 ;;
esac
