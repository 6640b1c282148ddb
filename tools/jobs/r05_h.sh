# Round-5 (h): bench A/B of the 256-channel Winograd stage (alternating pairs, one box)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/h; mkdir -p $O
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['utterances_per_sec'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'])" || tail -3 $O/b_$tag.err; }
run w1 SS_CONV_C256_WINOGRAD=1
run d1 SS_CONV_C256_WINOGRAD=0
run w2 SS_CONV_C256_WINOGRAD=1
run d2 SS_CONV_C256_WINOGRAD=0
run w3 SS_CONV_C256_WINOGRAD=1
run d3 SS_CONV_C256_WINOGRAD=0
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_batch_gpu.py -q -m gpu -k "c256 or vocoder" > $O/tests.log 2>&1; tail -4 $O/tests.log
