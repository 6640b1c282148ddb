# Round-4: one-stream kernel statistics of the bench workload (rocprofv3 --kernel-trace --stats) + share table
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/prof1; mkdir -p $O
export TMPDIR=/tmp
X="--no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1stream -- python bench.py --steps 20 --warmup 5 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X > $O/bench_1stream_under_rocprof.json 2> $O/prof_1stream.err
rm -f $O/*/*/*kernel_trace.csv
python tools/share_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/bench_1stream_under_rocprof.json > $O/share_table.md; cat $O/share_table.md
head -40 "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" | cut -c1-200
