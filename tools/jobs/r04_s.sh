# Round-4 (s): larger ragged batches: 64 / 96 / 128 utterances per pack at the driver's --steps 20
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04/s; mkdir -p $O
X="--gpus 1 --warmup 5 --no-cpu-baseline --no-multilingual --no-bf16x3-line --no-streaming-line --no-rccl-probe --no-bracket-ab --no-latency-pass"
for cfg in "64 20" "96 20" "128 20" "96 14" "128 10"; do
set -- $cfg
timeout 900 python bench.py $X --batch $1 --steps $2 > $O/b_$1_$2.json 2> $O/bench_$1.err; python -c "import json; d=json.load(open('$O/b_$1_$2.json')); print('batch $1 x $2 steps:', d['value'], d['utterances_per_sec'], d['ms_per_step'], d['roofline']['frac'])" || tail -3 $O/bench_$1.err
done
