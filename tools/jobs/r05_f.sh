# Round-5 (f): the Winograd slab kernel at 256 channels -- op tests, micro-benchmark against conv_sk2<128>, bench A/B
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/f; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "c256 or c128" > $O/ops.log 2>&1; tail -8 $O/ops.log
C64_BENCH_CHANNELS=256 timeout 300 python tools/c64_bench.py > $O/c256_micro.txt 2>&1; cat $O/c256_micro.txt
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['utterances_per_sec'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])" || tail -3 $O/b_$tag.err; }
run w1 SS_CONV_C256_WINOGRAD=1
run d1 SS_CONV_C256_WINOGRAD=0
run w2 SS_CONV_C256_WINOGRAD=1
run d2 SS_CONV_C256_WINOGRAD=0
