# Round-5 (m): one-stream kernel statistics + share table of the current build; per-conv Winograd from k = 7 at 32 channels (A/B)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/m; mkdir -p $O
export TMPDIR=/tmp
X="--no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe --no-soak"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1stream -- python bench.py --steps 20 --warmup 5 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X > $O/bench_1stream_under_rocprof.json 2> $O/prof_1stream.err
rm -f $O/*/*/*kernel_trace.csv
python tools/share_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/bench_1stream_under_rocprof.json > $O/share_table.md; cat $O/share_table.md
XX="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $XX > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['ms_per_step'])" || tail -3 $O/b_$tag.err; }
run k11a A=1
run k7a SS_CONV_C32_MIN_K=7
run k11b A=1
run k7b SS_CONV_C32_MIN_K=7
run k3a SS_CONV_C32_MIN_K=3
run k11c A=1
