# Round-4 (ao): Winograd form at 32 channels: op tests, micro-benchmark (slab columns) with the form off / on, bench A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/ao; mkdir -p $O
( timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "conv_c64_slab_kernel" ) > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for w in 0 1; do
  echo "== SS_CONV_C32_WINOGRAD=$w"
  SS_CONV_C32_WINOGRAD=$w C64_BENCH_CHANNELS=32 timeout 200 python tools/c64_bench.py 2>&1 | grep -E "^ +(3|7|11) +(1|5) " | awk -F'|' '{print $1 "|" $3 "|" $5}'
done | tee $O/micro32.txt
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-prof"
for w in 0 1 0 1; do
  SS_CONV_C32_WINOGRAD=$w timeout 600 python bench.py $X > $O/b_$w.json 2> $O/b_$w.err; python -c "import json; d=json.load(open('$O/b_$w.json')); print('c32 winograd $w:', d['value'], d['ms_per_step'])" || tail -3 $O/b_$w.err
done 2>&1 | tee $O/bench_ab.txt
