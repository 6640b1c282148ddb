# Round-5 (z), packs of 128: occupancy in time of the 8-stream timed region (rocprofv3 kernel trace + tools/trace_gaps.py), and of one stream
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/z; mkdir -p $O
export TMPDIR=/tmp
X="--steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak --no-prof"
for try in 1 2 3; do
  rm -rf $O/trace8
  SS_BENCH_NO_REPLAY=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace8 -- python bench.py $X > $O/bench8.json 2> $O/trace8.err && ls $O/trace8/*/*kernel_trace.csv > /dev/null 2>&1 && break
  echo "try $try failed"
done
python tools/trace_gaps.py "$(ls $O/trace8/*/*kernel_trace.csv | head -1)" $O/bench8.json > $O/trace_gaps_8streams.txt 2>&1; cat $O/trace_gaps_8streams.txt
SS_BENCH_NO_REPLAY=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python bench.py $X --streams 1 > $O/bench1.json 2> $O/trace1.err
python tools/trace_gaps.py "$(ls $O/trace1/*/*kernel_trace.csv | head -1)" $O/bench1.json > $O/trace_gaps_1stream.txt 2>&1; cat $O/trace_gaps_1stream.txt
python -c "import json; a=json.load(open('$O/bench8.json')); b=json.load(open('$O/bench1.json')); print('8 streams', a['value'], a['ms_per_step'], ' 1 stream', b['value'], b['ms_per_step'])"
rm -f $O/trace*/*/*kernel_trace.csv
# stream count at packs of 128 (memory: 14.9 GB of scratch per stream)
X2="--steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
for s in 4 6 8 12; do
  timeout 300 python bench.py $X2 --streams $s > $O/b_s$s.json 2> $O/b_s$s.err; python -c "import json; d=json.load(open('$O/b_s$s.json')); print('streams $s:', d['value'], d['ms_per_step'], d['hbm']['in_use_after_the_timed_region_gb'])" || tail -2 $O/b_s$s.err
done
