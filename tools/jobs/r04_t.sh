# Round-4 (t): bench default -> 64 utterances per ragged batch: the two tests that pin the measured configuration, then the
# per-conv thresholds of the narrow vocoder stages re-checked at that pack size (they were set at 32)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/t; mkdir -p $O
timeout 900 python -m pytest tests/test_bench_config_gpu.py tests/test_margin_gpu.py -x -q -m gpu -s 2>&1 | tail -15 > $O/tests.log; cat $O/tests.log
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['utterances_per_sec'], d['ms_per_step'], d['roofline']['frac'], d['config']['utterances_per_ragged_batch'])" || tail -3 $O/b_$tag.err; }
run default A=1
run c32k3 SS_CONV_C32_MIN_K=3
run c16k3 SS_CONV_C16_MIN_K=3
run c32c16k3 SS_CONV_C32_MIN_K=3 SS_CONV_C16_MIN_K=3
run c32k7 SS_CONV_C32_MIN_K=7 SS_CONV_C16_MIN_K=7
run default2 A=1
