# Round-4 (ak): rocprofv3 --kernel-trace --stats around the driver's 8-stream command (its own interception segfaults in most runs: up to 4 tries)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/ak; mkdir -p $O
export TMPDIR=/tmp
for try in 1 2 3 4; do
  rm -rf $O/prof_driver
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streaming-line --no-multilingual --no-bf16x3-line --no-rccl-probe > $O/bench_under_rocprof.json 2> $O/prof_driver.err && ls $O/prof_driver/*/*_kernel_stats.csv > /dev/null 2>&1 && { echo "try $try ok"; break; }
  echo "try $try failed"
done
rm -f $O/prof_driver/*/*kernel_trace.csv
