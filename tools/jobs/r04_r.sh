# Round-4 (r): ragged-batch size sweep (same 640-1280 utterances): 32 (default) vs 48 / 64 per batch
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/r; mkdir -p $O
X="--gpus 1 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for cfg in "32 40" "64 20" "48 27" "32 40" "64 20"; do
set -- $cfg
timeout 900 python bench.py $X --batch $1 --steps $2 > $O/b_$1.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/b_$1.json')); print('batch $1 x $2 steps:', d['value'], d['utterances_per_sec'], d['ms_per_step'], d['roofline']['frac'])"
done
