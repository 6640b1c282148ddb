# Round-4 (u): knobs that were set at 32 utterances per pack, re-checked at the new default of 64 (64 x 20 steps)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/u; mkdir -p $O
X="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py $X $EXTRA > $O/b_$tag.json 2> $O/b_$tag.err; python -c "import json; d=json.load(open('$O/b_$tag.json')); print('$tag:', d['value'], d['utterances_per_sec'], d['ms_per_step'], d['roofline']['frac'])" || tail -3 $O/b_$tag.err; }
run default A=1
EXTRA="--streams 6" run streams6 A=1
EXTRA="--streams 10" run streams10 A=1
EXTRA="--streams 12" run streams12 A=1
EXTRA=""
run ffn_wm4 SS_FFN_WM=4
run rtlin_units2000 SS_RTLIN_MIN_UNITS=2000
run c64_wm3 SS_CONV_C64_WM=3
run sk_spare SS_SK2_SPARE_CUS=8
run default2 A=1
