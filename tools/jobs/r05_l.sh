# Round-5 (l): k = 7 at dilation 5 on the 64-channel Winograd kernel, now that a 7-tap conv costs 10 instead of 12 k-blocks
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05/l; mkdir -p $O
SS_CONV_C64_WINOGRAD_K7D5=1 C64_BENCH_CHANNELS=64 timeout 300 python tools/c64_bench.py > $O/micro_c64_k7d5.txt 2>&1; grep "^   7" $O/micro_c64_k7d5.txt
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass --no-soak"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['ms_per_step'])" || tail -3 $O/b_$tag.err; }
run a1 SS_CONV_C64_WINOGRAD_K7D5=1
run b1 SS_CONV_C64_WINOGRAD_K7D5=0
run a2 SS_CONV_C64_WINOGRAD_K7D5=1
run b2 SS_CONV_C64_WINOGRAD_K7D5=0
