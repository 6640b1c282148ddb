# Round-4 final: GPU suite + smoke, then the evidence passes of the committed build without the (flaky) rocprofv3 run around the 8-stream command
cd "$GRAFT_REPO_ROOT"
bash tools/jobs/r04_w.sh
O=gpurun_out/r04/full
mkdir -p $O
export TMPDIR=/tmp
X="--no-multilingual --no-streaming-line --no-bracket-ab --no-rccl-probe"
rm -rf $O/prof_1stream $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_MfmaUtil
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_1stream -- python bench.py --steps 20 --warmup 5 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X > $O/bench_1stream_under_rocprof.json 2> $O/prof_1stream.err
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python bench.py --steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X > $O/pmc_$c.bench.json 2> $O/pmc_$c.err
done
F=$(ls $O/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1); U=$(ls $O/pmc_MfmaUtil/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $F $W $O/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc MfmaUtil (three separate passes, --kernel-trace only) of bench.py --steps 6 --warmup 1 --streams 1 --no-latency-pass --no-cpu-baseline --no-bf16x3-line $X; gfx950 correction: HBM read bytes = 2*FETCH_SIZE KB, WRITE_SIZE as reported; algorithmic bytes = the library's census of the same process" $U $O/pmc_FETCH_SIZE.bench.json > $O/pmc_classes.txt 2>&1
rm -f $O/pmc_*/*/*counter_collection.csv $O/*/*/*kernel_trace.csv
python tools/share_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/bench_1stream_under_rocprof.json > $O/share_table.md; head -14 $O/share_table.md
python tools/roofline_table.py "$(ls -t $O/prof_1stream/*/*_kernel_stats.csv | head -1)" $O/pmc_traffic.json > $O/roofline_table.md
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json     # so that the bench line below joins its traffic to counters of THIS build
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_detail']['same_kernel_sources'], d['roofline_second_kernel']['kernel'], d['roofline_second_kernel']['frac'])"
